#!/bin/bash
# the front-end's per-stream glue kernels as one wavefront per stream: parity (forced at the tests' small batches), front-end alone, complete path
set -o pipefail
O=$PWD/gpurun_out/r05q; mkdir -p $O
AV_FE_GLUE_WG=64 timeout -k 10 900 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_stages.py -x -q -m gpu > $O/pytest_glue.txt 2>&1 || { tail -30 $O/pytest_glue.txt; exit 1; }
tail -1 $O/pytest_glue.txt
for v in "AV_X=0" "AV_FE_GLUE_WG=256"; do
env $v python bench.py --frontend-only --no-cpu-baseline > $O/bench_r.json 2> $O/e1.txt || { tail -5 $O/e1.txt; exit 1; }
python3 - $O/bench_r.json "$v fe-only" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print('%-26s value %.0f ms/step %.2f kernels %s' % (sys.argv[2], d['value'], d['ms_per_step'], json.dumps({k: round(v,2) for k,v in d.get('kernel_ms_per_step',{}).items()})))
PY
env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_r.json 2> $O/e1.txt || { tail -5 $O/e1.txt; exit 1; }
python3 - $O/bench_r.json "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); m=d.get('roofline_msckf') or {}
print('%-26s value %.0f ms/step %.2f  chain %.2f  kernels %s' % (sys.argv[2], d['value'], d['ms_per_step'], m.get('chain_ms_per_step'), json.dumps({k: round(v,2) for k,v in d.get('kernel_ms_per_step',{}).items()})))
PY
done
