#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/r05q; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_msckf.py -x -q -m gpu -k "golden or oracle or blank" > $O/pytest_msckf.txt 2>&1 || { tail -30 $O/pytest_msckf.txt; exit 1; }
tail -2 $O/pytest_msckf.txt
for v in "AV_X=0" "AV_FEAT_W64_MAX=4" "AV_X=1" "AV_FEAT_W64_MAX=12"; do
env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_n.json 2> $O/e1.txt || { tail -5 $O/e1.txt; exit 1; }
python3 - $O/bench_n.json "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); m=d.get('roofline_msckf') or {}
print('%-20s value %.0f ms/step %.2f  msckf chain %.2f excl %s  kernels %s' % (sys.argv[2], d['value'], d['ms_per_step'], m.get('chain_ms_per_step'), (m.get('exclusive') or {}).get('chain_ms_per_step'), json.dumps({k: round(v,2) for k,v in d.get('kernel_ms_per_step',{}).items()})))
PY
done
