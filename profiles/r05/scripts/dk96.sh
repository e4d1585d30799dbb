#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/r05q; mkdir -p $O
for v in "AV_X=0" "AV_X=1" "AV_X=2"; do
env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_m.json 2> $O/e1.txt || { tail -5 $O/e1.txt; exit 1; }
python3 - $O/bench_m.json "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); m=d.get('roofline_msckf') or {}
print('%-14s value %.0f ms/step %.2f  msckf chain %.2f excl %s  kernels %s' % (sys.argv[2], d['value'], d['ms_per_step'], m.get('chain_ms_per_step'), (m.get('exclusive') or {}).get('chain_ms_per_step'), json.dumps({k: round(v,2) for k,v in d.get('kernel_ms_per_step',{}).items()})))
PY
done
