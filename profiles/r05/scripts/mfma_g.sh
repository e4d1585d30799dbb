#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/r05q; mkdir -p $O
for v in "AV_X=0" "AV_MSCKF_PRIO=2" "AV_MSCKF_GROUPS=1" "AV_MSCKF_GROUPS=3" "AV_MSCKF_GROUPS=4 GPU_MAX_HW_QUEUES=8"; do
env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_h.json 2> $O/e1.txt || { tail -5 $O/e1.txt; exit 1; }
python3 - $O/bench_h.json "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); m=d.get('roofline_msckf') or {}
print('%-40s value %.0f ms/step %.2f  msckf chain %.2f excl %s  kernels %s' % (sys.argv[2], d['value'], d['ms_per_step'], m.get('chain_ms_per_step'), (m.get('exclusive') or {}).get('chain_ms_per_step'), json.dumps({k: round(v,2) for k,v in d.get('kernel_ms_per_step',{}).items()})))
PY
done
bash profiles/r05/scripts/contended_timeline.sh && cat gpurun_out/r05y/contended_timeline.txt
