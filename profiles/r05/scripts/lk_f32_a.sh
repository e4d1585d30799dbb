#!/bin/bash
# LK without fp64 in the Newton loop + the fixed 48-wide update tiles: microbench, parity tests, A/B bench lines, LK phase stamps
set -o pipefail
O=$PWD/gpurun_out/r05c; mkdir -p $O
./profiles/r05/valu_issue_microbench > $O/valu_issue_microbench.json 2> $O/mb.err || { tail -5 $O/mb.err; exit 1; }
python3 - $O/valu_issue_microbench.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for r in d['rows']: print("%-28s w1 %6.2f  w2 %6.2f  w4 %6.2f  w5 %6.2f  w5 %.3f ns  ticks/us %s" % (r["op"], r["w1"], r["w2"], r["w4"], r["w5"], r["w5_ns"], r.get("ticks_per_us")))
PY
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_frontend.py tests/test_gpu_msckf.py -m gpu -x -q > $O/pytest_subset.txt 2>&1; rc=$?; tail -5 $O/pytest_subset.txt; [ $rc -eq 0 ] || exit 1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_tile48.json 2> $O/e_tile48.txt; echo "tile48 rc $?"; tail -2 $O/e_tile48.txt
AV_UPD_TILE=64 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_tile64.json 2> $O/e_tile64.txt; echo "tile64 rc $?"; tail -2 $O/e_tile64.txt
AV_LK_PROF=1 python bench.py --frontend-only --steps 10 --warmup 3 --no-cpu-baseline --no-regimes > $O/bench_fe_prof.json 2> $O/lk_phase_stamps.txt; echo "prof rc $?"; grep AV_LK_PROF $O/lk_phase_stamps.txt
for f in bench_tile48 bench_tile64; do python3 - $O/$f.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline']
    print(sys.argv[1].split('/')[-1], 'value %.0f ms/step %.2f fe_only %s lk_ms %.3f (alone %s) frac %.4f kernels %s excl %s' % (d['value'], d['ms_per_step'], d.get('frontend_only_frames_per_s'), r['avg_launch_ms'], r.get('avg_launch_ms_frontend_only'), r['frac'], json.dumps(d.get('kernel_ms_per_step')), json.dumps((d.get('roofline_msckf') or {}).get('exclusive'))[:300]))
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY
done
