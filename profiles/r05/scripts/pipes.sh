#!/bin/bash
# independent pipelines per GPU (bench.py --pipelines P)
set -o pipefail
O=$PWD/gpurun_out/r05q; mkdir -p $O
for v in "--pipelines 2" "--pipelines 4" "--pipelines 4 X AV_MSCKF_GROUPS=1" "--pipelines 3"; do
a=${v%% X *}; e=${v##* X }; [ "$e" = "$v" ] && e="AV_X=0"
env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes $a > $O/bench_t.json 2> $O/e1.txt || { tail -8 $O/e1.txt; exit 1; }
python3 - $O/bench_t.json "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); m=d.get('roofline_msckf') or {}; r=d['roofline']
print('%-36s value %.0f ms/step %.2f fe_only %.0f lk_ms %.3f frac %.4f chain %.2f kernels %s ss %s' % (sys.argv[2], d['value'], d['ms_per_step'], d.get('frontend_only_frames_per_s') or 0, r['avg_launch_ms'], r['frac'], m.get('chain_ms_per_step'), json.dumps({k: round(v,2) for k,v in d.get('kernel_ms_per_step',{}).items()}), d['steady_state']['cam_states_at_t0']))
PY
done
