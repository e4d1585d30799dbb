#!/bin/bash
# PMC passes of the front-end at 64 and at 512 streams per launch (SQ block, FETCH_SIZE, WRITE_SIZE: three separate passes each)
set -o pipefail
O=gpurun_out/r05_pmc; mkdir -p $O
bash profiles/r05/collect_pmc.sh $O/s64 --frontend-only --streams 64 --steps 6 --warmup 2 > $O/s64.log 2>&1 || { tail -20 $O/s64.log; exit 1; }
cp $O/s64/pmc_summary.json $O/pmc_frontend_s64_summary.json
bash profiles/r05/collect_pmc.sh $O/s512 --frontend-only --streams 512 --steps 4 --warmup 2 > $O/s512.log 2>&1 || { tail -20 $O/s512.log; exit 1; }
cp $O/s512/pmc_summary.json $O/pmc_frontend_s512_summary.json
python3 - <<'PY'
import json
for s in (64, 512):
    d=json.load(open('gpurun_out/r05_pmc/pmc_frontend_s%d_summary.json' % s))
    for k in ('lk_track_g16_kernel<15>','fast_kernel','pyr_l0l1_kernel<false, 96>','pyr_l2l3_kernel<1024>','select_kernel'):
        e=d.get(k)
        if e: print(s, k, {c: round(e[c],3) for c in ('SQ_INSTS_VALU','SQ_WAVES','valu_per_wave','FETCH_SIZE','WRITE_SIZE','wait_any_share','wait_inst_share','dispatches') if c in e})
PY
