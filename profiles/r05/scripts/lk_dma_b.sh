#!/bin/bash
# LDS-DMA staged LK, pieces batched per stage: parity, then A/B of 5-wave DMA / 6-wave DMA / register-staged (front-end only x2 each, complete path)
set -o pipefail
O=$PWD/gpurun_out/r05f; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_frontend.py -m gpu -x -q > $O/pytest_dma5.txt 2>&1; rc=$?; tail -3 $O/pytest_dma5.txt; [ $rc -eq 0 ] || exit 1
AV_LK_DMA=6 timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_frontend.py -m gpu -x -q > $O/pytest_dma6.txt 2>&1; rc=$?; tail -3 $O/pytest_dma6.txt; [ $rc -eq 0 ] || exit 1
for rep in 1 2; do for dma in 1 6 0; do
AV_LK_DMA=$dma python bench.py --frontend-only --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_fe_dma${dma}_$rep.json 2> $O/e.txt; echo "fe dma=$dma rc $?"
done; done
for dma in 1 6 0; do
AV_LK_DMA=$dma python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_dma${dma}.json 2> $O/e.txt; echo "full dma=$dma rc $?"
done
for f in bench_fe_dma1_1 bench_fe_dma6_1 bench_fe_dma0_1 bench_fe_dma1_2 bench_fe_dma6_2 bench_fe_dma0_2 bench_dma1 bench_dma6 bench_dma0; do python3 - $O/$f.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline']
    print(sys.argv[1].split('/')[-1], 'value %.0f ms/step %.2f fe_only %s lk_ms %.3f (alone %s) frac %.4f kernels %s' % (d['value'], d['ms_per_step'], d.get('frontend_only_frames_per_s'), r['avg_launch_ms'], r.get('avg_launch_ms_frontend_only'), r['frac'], json.dumps(d.get('kernel_ms_per_step'))))
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY
done
