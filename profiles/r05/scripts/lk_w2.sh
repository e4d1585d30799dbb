#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/r05q; mkdir -p $O
for v in "AV_LK_WG=128" "AV_LK_WG=64"; do
env $v python bench.py --frontend-only --no-cpu-baseline > $O/bench_g.json 2> $O/e1.txt || { tail -5 $O/e1.txt; exit 1; }
python3 - $O/bench_g.json "$v fe-only" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print('%-30s value %.0f ms/step %.2f lk_ms %.3f kernels %s' % (sys.argv[2], d['value'], d['ms_per_step'], r['avg_launch_ms'], json.dumps({k: round(v,2) for k,v in d.get('kernel_ms_per_step',{}).items()})))
PY
env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_g.json 2> $O/e1.txt || { tail -5 $O/e1.txt; exit 1; }
python3 - $O/bench_g.json "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); m=d.get('roofline_msckf') or {}
print('%-30s value %.0f ms/step %.2f  msckf chain %.2f excl %s  kernels %s' % (sys.argv[2], d['value'], d['ms_per_step'], m.get('chain_ms_per_step'), (m.get('exclusive') or {}).get('chain_ms_per_step'), json.dumps({k: round(v,2) for k,v in d.get('kernel_ms_per_step',{}).items()})))
PY
done
