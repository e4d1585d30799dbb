#!/bin/bash
# lost-feature rows without the zero fill (masked readers): filter parity, then A/B against the cleared rows
set -o pipefail
O=$PWD/gpurun_out/r05k; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_msckf.py tests/test_gpu_sweep.py -m gpu -x -q > $O/pytest_msckf.txt 2>&1; rc=$?; tail -4 $O/pytest_msckf.txt; [ $rc -eq 0 ] || exit 1
for zf in 0 1 0 1; do
AV_MSCKF_ZERO_FILL=$zf python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_zf${zf}_$RANDOM.json 2> $O/e.txt; echo "zero_fill=$zf rc $?"
done
for f in $O/bench_zf*.json; do python3 - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); m=d['roofline_msckf']
print(sys.argv[1].split('/')[-1], 'value %.0f ms/step %.2f chain %.2f exclusive chain %.3f' % (d['value'], d['ms_per_step'], m['chain_ms_per_step'], m['exclusive']['chain_ms_per_step']))
PY
done
