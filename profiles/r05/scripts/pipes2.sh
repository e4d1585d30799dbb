#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/r05q; mkdir -p $O
for v in "--pipelines 2 X AV_MSCKF_GROUPS=1" "--pipelines 4 X AV_MSCKF_GROUPS=1 AV_DK_WG=64" "--pipelines 2 X AV_MSCKF_GROUPS=1 AV_BENCH_PRESTAGE=0"; do
a=${v%% X *}; e=${v##* X }
env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes $a > $O/bench_u.json 2> $O/e1.txt || { tail -8 $O/e1.txt; exit 1; }
python3 - $O/bench_u.json "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); m=d.get('roofline_msckf') or {}; r=d['roofline']
print('%-60s value %.0f ms/step %.2f fe_only %.0f chain %.2f kernels %s' % (sys.argv[2], d['value'], d['ms_per_step'], d.get('frontend_only_frames_per_s') or 0, m.get('chain_ms_per_step'), json.dumps({k: round(v,2) for k,v in d.get('kernel_ms_per_step',{}).items()})))
PY
done
python bench.py --frontend-only --no-cpu-baseline --pipelines 4 > $O/bench_frontend_only_p4.json 2> $O/e2.txt && python3 -c "
import json; d=json.load(open('$O/bench_frontend_only_p4.json')); print('fe-only p4: %.0f frames/s %.2f ms/step' % (d['value'], d['ms_per_step']))"
