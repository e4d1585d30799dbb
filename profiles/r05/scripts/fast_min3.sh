#!/bin/bash
# FAST score / NMS with v_pk_minimum3_f16 / v_pk_maximum3_f16: parity, then front-end only and complete path
set -o pipefail
O=$PWD/gpurun_out/r05q; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_frontend.py -m gpu -x -q > $O/pytest_subset.txt 2>&1; rc=$?; tail -3 $O/pytest_subset.txt; [ $rc -eq 0 ] || exit 1
for rep in 1 2; do
python bench.py --frontend-only --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_fe_$rep.json 2> $O/e.txt; echo "fe rc $?"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_full_$rep.json 2> $O/e.txt; echo "full rc $?"
done
for f in $O/bench_fe_*.json $O/bench_full_*.json; do python3 - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[1].split('/')[-1], 'value %.0f ms/step %.2f fe_only %s lk_ms %.3f kernels %s' % (d['value'], d['ms_per_step'], d.get('frontend_only_frames_per_s'), r['avg_launch_ms'], json.dumps(d.get('kernel_ms_per_step'))))
PY
done
