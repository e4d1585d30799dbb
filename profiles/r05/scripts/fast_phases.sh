#!/bin/bash
# what the detector's phases cost (AV_FAST_DBG knock-outs: 1 no score pass, 2 no compass test (nothing survives), 4 no NMS, 8 no output; results are wrong)
O=$PWD/gpurun_out/r05o; mkdir -p $O
for dbg in 0 1 2 4 8; do
AV_FAST_DBG=$dbg python bench.py --frontend-only --steps 10 --warmup 3 --no-cpu-baseline --no-regimes > $O/bench_fe_dbg$dbg.json 2> $O/e.txt
python3 -c "
import json; d=json.load(open('$O/bench_fe_dbg$dbg.json')); print('AV_FAST_DBG=$dbg fast %.3f ms per step, glue %.3f, lk %.3f' % (d['kernel_ms_per_step']['fast'], d['kernel_ms_per_step']['glue'], d['kernel_ms_per_step']['lk']))"
done
