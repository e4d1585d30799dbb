#!/bin/bash
# what the driver runs at round end: the GPU suite and smoke()
set -o pipefail
O=$PWD/gpurun_out/r05_check; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; tail -4 $O/pytest_gpu.txt; [ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
