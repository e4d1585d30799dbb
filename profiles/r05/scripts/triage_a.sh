#!/bin/bash
# which of the round's unmeasured changes breaks the pre-roll: the 48-wide tile kernels (AV_UPD_TILE=64 switches them off)?
set -o pipefail
O=$PWD/gpurun_out/r05b; mkdir -p $O
AV_UPD_TILE=64 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_tile64.json 2> $O/e_tile64.txt; echo "tile64 rc $?"; tail -2 $O/e_tile64.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_tile48.json 2> $O/e_tile48.txt; echo "tile48 rc $?"; tail -2 $O/e_tile48.txt
AV_UPD_TILE=64 timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_gpu_tile64.txt 2>&1; echo "pytest tile64 exit $?"; tail -15 $O/pytest_gpu_tile64.txt
