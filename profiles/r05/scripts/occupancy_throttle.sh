#!/bin/bash
# do the occupancy-saturating front-end kernels (FAST: 8 workgroups per CU, pyramid: 6) starve the filter's queues?  LDS-pad A/B, complete path
set -o pipefail
O=$PWD/gpurun_out/r05i; mkdir -p $O
run() { # name, env...
  local name=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_$name.json 2> $O/e_$name.txt
  python3 - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline']
    print('%-22s value %.0f ms/step %.2f fe_only %.0f kernels %s chain_ms %.2f' % (sys.argv[2], d['value'], d['ms_per_step'], d.get('frontend_only_frames_per_s') or 0, json.dumps(d.get('kernel_ms_per_step')), d['roofline_msckf']['chain_ms_per_step']))
except Exception as e: print(sys.argv[2], 'unreadable', e)
PY
}
run base AV_X=0
run fast6 AV_FAST_LDS_PAD=6000
run fast5 AV_FAST_LDS_PAD=12000
run fast4 AV_FAST_LDS_PAD=20000
run pyr5 AV_PYR_LDS_PAD=3000
run pyr4 AV_PYR_LDS_PAD=9000
run pyr3 AV_PYR_LDS_PAD=18000
run fast5pyr4 AV_FAST_LDS_PAD=12000 AV_PYR_LDS_PAD=9000
run base2 AV_X=0
run feat8_v128 AV_FILTER_V128=1
run info_v128 AV_FILTER_V128=2
run both_v128 AV_FILTER_V128=3
run base3 AV_X=0
