#!/bin/bash
# LDS-DMA staged LK: parity (ops + front-end engine), then A/B front-end-only and complete path, stamps of both kernels
set -o pipefail
O=$PWD/gpurun_out/r05e; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_frontend.py -m gpu -x -q > $O/pytest_subset.txt 2>&1; rc=$?; tail -5 $O/pytest_subset.txt; [ $rc -eq 0 ] || exit 1
for dma in 1 0; do
AV_LK_DMA=$dma python bench.py --frontend-only --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_fe_dma$dma.json 2> $O/e_fe_dma$dma.txt; echo "fe dma=$dma rc $?"
AV_LK_DMA=$dma python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_dma$dma.json 2> $O/e_dma$dma.txt; echo "full dma=$dma rc $?"
AV_LK_DMA=$dma AV_LK_PROF=1 python bench.py --frontend-only --steps 10 --warmup 3 --no-cpu-baseline --no-regimes > $O/bench_fe_prof_dma$dma.json 2> $O/lk_phase_stamps_dma$dma.txt; grep AV_LK_PROF $O/lk_phase_stamps_dma$dma.txt
done
for f in bench_fe_dma1 bench_fe_dma0 bench_dma1 bench_dma0; do python3 - $O/$f.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline']
    print(sys.argv[1].split('/')[-1], 'value %.0f ms/step %.2f fe_only %s lk_ms %.3f (alone %s) frac %.4f kernels %s' % (d['value'], d['ms_per_step'], d.get('frontend_only_frames_per_s'), r['avg_launch_ms'], r.get('avg_launch_ms_frontend_only'), r['frac'], json.dumps(d.get('kernel_ms_per_step'))))
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY
done
