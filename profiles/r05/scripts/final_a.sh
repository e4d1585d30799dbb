#!/bin/bash
# round-5 final bench lines: driver command, default, front-end only, configs[4] shape
set -o pipefail
O=$PWD/gpurun_out/r05_final; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd_s2048_r05.json 2> $O/e1.txt || { tail -5 $O/e1.txt; exit 1; }
python bench.py > $O/bench_default_s2048_r05.json 2> $O/e2.txt || { tail -5 $O/e2.txt; exit 1; }
python bench.py --frontend-only --no-cpu-baseline > $O/bench_frontend_only_s2048_r05.json 2> $O/e3.txt || { tail -5 $O/e3.txt; exit 1; }
python bench.py --grid 10 15 10 --streams 256 --steps 20 --warmup 5 > $O/bench_config5_grid10x15x10_s256_r05.json 2> $O/e4.txt || { tail -5 $O/e4.txt; exit 1; }
for f in bench_driver_cmd_s2048_r05 bench_default_s2048_r05 bench_frontend_only_s2048_r05 bench_config5_grid10x15x10_s256_r05; do python3 - $O/$f.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']; m=d.get('roofline_msckf') or {}
print(sys.argv[1].split('/')[-1], 'value %.0f ms/step %.2f fe_only %s lk_ms %.3f (alone %s) frac %.4f traffic %.0f valu %s msckf %s excl %s executed %s' % (d['value'], d['ms_per_step'], d.get('frontend_only_frames_per_s'), r['avg_launch_ms'], r.get('avg_launch_ms_frontend_only'), r['frac'], r['traffic'], r.get('valu_issue_frac'), m.get('frac'), (m.get('exclusive') or {}).get('frac'), json.dumps(m.get('executed'))))
print('   kernels', json.dumps(d.get('kernel_ms_per_step')), 'cpu', json.dumps(d.get('cpu_baseline'))[:200], 'ate', json.dumps(d.get('ate_vs_cpu_ref'))[:160])
if d.get('regimes'): print('   regimes', [(g['name'], round(g['ms_per_step'],3), round(g['stream_frames_per_s'])) for g in d['regimes']])
PY
done
