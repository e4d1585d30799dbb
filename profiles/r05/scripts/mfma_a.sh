#!/bin/bash
# MFMA back end: filter parity tests, then the exclusive timeline of the filter chain and the driver command
set -o pipefail
O=$PWD/gpurun_out/r05q; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_msckf.py -x -q -m gpu > $O/pytest_msckf.txt 2>&1 || { tail -30 $O/pytest_msckf.txt; exit 1; }
tail -3 $O/pytest_msckf.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_mfma.json 2> $O/e1.txt || { tail -5 $O/e1.txt; exit 1; }
python3 - $O/bench_mfma.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); m=d.get('roofline_msckf') or {}
print('value %.0f ms/step %.2f  msckf chain %.2f excl %s' % (d['value'], d['ms_per_step'], m.get('chain_ms_per_step'), (m.get('exclusive') or {}).get('chain_ms_per_step')))
PY
bash profiles/r05/scripts/exclusive_timeline.sh
