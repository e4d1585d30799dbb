#!/bin/bash
# store kernels (dk_begin4 / dk_mid / dk_finish) as one wavefront per stream: parity, A/B, timelines
set -o pipefail
O=$PWD/gpurun_out/r05q; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_msckf.py tests/test_gpu_pipeline.py -x -q -m gpu > $O/pytest_msckf.txt 2>&1 || { tail -30 $O/pytest_msckf.txt; exit 1; }
tail -2 $O/pytest_msckf.txt
for v in "AV_X=0" "AV_DK_WG=256" "AV_X=1"; do
env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_l.json 2> $O/e1.txt || { tail -5 $O/e1.txt; exit 1; }
python3 - $O/bench_l.json "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); m=d.get('roofline_msckf') or {}
print('%-14s value %.0f ms/step %.2f  msckf chain %.2f excl %s  kernels %s' % (sys.argv[2], d['value'], d['ms_per_step'], m.get('chain_ms_per_step'), (m.get('exclusive') or {}).get('chain_ms_per_step'), json.dumps({k: round(v,2) for k,v in d.get('kernel_ms_per_step',{}).items()})))
PY
done
bash profiles/r05/scripts/exclusive_timeline.sh > $O/excl.txt 2>&1; grep "^dk_\|^upd_\|^feature\|^triang" $O/excl.txt | grep start
bash profiles/r05/scripts/contended_timeline.sh && grep "^period\|^front-end\|^sum\|^dk_\|^pyr\|^feature_kernel<\|^upd_info\|^fast" gpurun_out/r05y/contended_timeline.txt | head -30
