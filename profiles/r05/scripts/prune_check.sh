#!/bin/bash
# after the pruning pass: the whole GPU suite, then A/B of FAST on a second stream and the six-per-CU pyramid tiles
set -o pipefail
O=$PWD/gpurun_out/r05g; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; tail -4 $O/pytest_gpu.txt; [ $rc -eq 0 ] || exit 1
for fa in 0 1; do
AV_FE_FAST_ASYNC=$fa python bench.py --frontend-only --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_fe_fa$fa.json 2> $O/e.txt; echo "fe fast_async=$fa rc $?"
AV_FE_FAST_ASYNC=$fa python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_fa$fa.json 2> $O/e.txt; echo "full fast_async=$fa rc $?"
done
for f in bench_fe_fa0 bench_fe_fa1 bench_fa0 bench_fa1; do python3 - $O/$f.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline']
    print(sys.argv[1].split('/')[-1], 'value %.0f ms/step %.2f fe_only %s lk_ms %.3f (alone %s) frac %.4f kernels %s' % (d['value'], d['ms_per_step'], d.get('frontend_only_frames_per_s'), r['avg_launch_ms'], r.get('avg_launch_ms_frontend_only'), r['frac'], json.dumps(d.get('kernel_ms_per_step'))))
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY
done
