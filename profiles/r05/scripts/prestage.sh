#!/bin/bash
# av_frontend_prestage: parity, then the driver command with / without
set -o pipefail
O=$PWD/gpurun_out/r05q; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_prestage.py -x -q -m gpu > $O/pytest_pre.txt 2>&1 || { tail -30 $O/pytest_pre.txt; exit 1; }
tail -2 $O/pytest_pre.txt
for v in "AV_X=0" "AV_BENCH_PRESTAGE=0" "AV_X=1"; do
env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_j.json 2> $O/e1.txt || { tail -5 $O/e1.txt; exit 1; }
python3 - $O/bench_j.json "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); m=d.get('roofline_msckf') or {}
print('%-24s value %.0f ms/step %.2f  msckf chain %.2f excl %s  kernels %s ate %s' % (sys.argv[2], d['value'], d['ms_per_step'], m.get('chain_ms_per_step'), (m.get('exclusive') or {}).get('chain_ms_per_step'), json.dumps({k: round(v,2) for k,v in d.get('kernel_ms_per_step',{}).items()}), d.get('steady_state')))
PY
done
bash profiles/r05/scripts/contended_timeline.sh && cat gpurun_out/r05y/contended_timeline.txt
