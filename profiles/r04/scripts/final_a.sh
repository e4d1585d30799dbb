#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/r04_final; mkdir -p $O
R=$GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd_s2048.json 2> $O/e1.txt || { tail -5 $O/e1.txt; exit 1; }
python bench.py > $O/bench_default_s2048.json 2> $O/e2.txt || { tail -5 $O/e2.txt; exit 1; }
python bench.py --frontend-only --no-cpu-baseline > $O/bench_frontend_only_s2048.json 2> $O/e3.txt || { tail -5 $O/e3.txt; exit 1; }
python bench.py --grid 10 15 10 --streams 256 --steps 20 --warmup 5 > $O/bench_config5_grid10x15x10_s256.json 2> $O/e4.txt || { tail -5 $O/e4.txt; exit 1; }
AV_MSCKF_STORE=host python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_driver_cmd_s2048_hoststore.json 2> $O/e5.txt || { tail -5 $O/e5.txt; exit 1; }
for f in bench_driver_cmd_s2048 bench_default_s2048 bench_frontend_only_s2048 bench_config5_grid10x15x10_s256 bench_driver_cmd_s2048_hoststore; do python3 - $O/$f.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[1].split('/')[-1], 'value %.0f ms/step %.2f fe_only %s lk_ms %.3f frac %.4f traffic %.0f msckf %s' % (d['value'], d['ms_per_step'], d.get('frontend_only_frames_per_s'), r['avg_launch_ms'], r['frac'], r['traffic'], (d.get('roofline_msckf') or {}).get('frac')))
PY
done
