#!/bin/bash
# sparse LK launch for the second matching round: parity, front-end alone, complete path
set -o pipefail
O=$PWD/gpurun_out/r05q; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_frontend.py -x -q -m gpu > $O/pytest_fe.txt 2>&1 || { tail -30 $O/pytest_fe.txt; exit 1; }
tail -1 $O/pytest_fe.txt
for v in "AV_X=0" "AV_X=1"; do
python bench.py --frontend-only --no-cpu-baseline > $O/bench_s.json 2> $O/e1.txt || { tail -5 $O/e1.txt; exit 1; }
python3 - $O/bench_s.json "$v fe-only" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print('%-16s value %.0f ms/step %.2f lk_ms %.3f kernels %s' % (sys.argv[2], d['value'], d['ms_per_step'], r['avg_launch_ms'], json.dumps({k: round(v,2) for k,v in d.get('kernel_ms_per_step',{}).items()})))
PY
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_s.json 2> $O/e1.txt || { tail -5 $O/e1.txt; exit 1; }
python3 - $O/bench_s.json "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); m=d.get('roofline_msckf') or {}
print('%-16s value %.0f ms/step %.2f  chain %.2f  kernels %s' % (sys.argv[2], d['value'], d['ms_per_step'], m.get('chain_ms_per_step'), json.dumps({k: round(v,2) for k,v in d.get('kernel_ms_per_step',{}).items()})))
PY
done
