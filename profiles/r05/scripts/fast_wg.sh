#!/bin/bash
# FAST tile worked by fewer wavefronts (AV_FAST_WG=128 / 64): parity, front-end alone, complete path
set -o pipefail
O=$PWD/gpurun_out/r05q; mkdir -p $O
for w in 128 64; do
AV_FAST_WG=$w timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_frontend.py -x -q -m gpu -k "fast or FAST or engine_matches_oracle_two" > $O/pytest_fast.txt 2>&1 || { tail -30 $O/pytest_fast.txt; exit 1; }
tail -1 $O/pytest_fast.txt
done
for v in "AV_FAST_WG=128" "AV_FAST_WG=64" "AV_FAST_WG=256"; do
env $v python bench.py --frontend-only --no-cpu-baseline > $O/bench_q.json 2> $O/e1.txt || { tail -5 $O/e1.txt; exit 1; }
python3 - $O/bench_q.json "$v fe-only" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print('%-24s value %.0f ms/step %.2f kernels %s' % (sys.argv[2], d['value'], d['ms_per_step'], json.dumps({k: round(v,2) for k,v in d.get('kernel_ms_per_step',{}).items()})))
PY
env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_q.json 2> $O/e1.txt || { tail -5 $O/e1.txt; exit 1; }
python3 - $O/bench_q.json "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); m=d.get('roofline_msckf') or {}
print('%-24s value %.0f ms/step %.2f  chain %.2f  kernels %s' % (sys.argv[2], d['value'], d['ms_per_step'], m.get('chain_ms_per_step'), json.dumps({k: round(v,2) for k,v in d.get('kernel_ms_per_step',{}).items()})))
PY
done
