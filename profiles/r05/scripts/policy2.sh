#!/bin/bash
# sharing policy once more, on the final chain
set -o pipefail
O=$PWD/gpurun_out/r05q; mkdir -p $O
for v in "AV_MSCKF_PRIO=2" "AV_MSCKF_GROUPS=1" "AV_MSCKF_GROUPS=4 GPU_MAX_HW_QUEUES=8" "AV_MSCKF_GROUPS=2 GPU_MAX_HW_QUEUES=8" "AV_MSCKF_GROUPS=4 GPU_MAX_HW_QUEUES=8 AV_MSCKF_PRIO=2 AV_DK_WG=64"; do
env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_p.json 2> $O/e1.txt || { tail -5 $O/e1.txt; exit 1; }
python3 - $O/bench_p.json "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); m=d.get('roofline_msckf') or {}
print('%-62s value %.0f ms/step %.2f  chain %.2f  kernels %s' % (sys.argv[2], d['value'], d['ms_per_step'], m.get('chain_ms_per_step'), json.dumps({k: round(v,2) for k,v in d.get('kernel_ms_per_step',{}).items()})))
PY
done
