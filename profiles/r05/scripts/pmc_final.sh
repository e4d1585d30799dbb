#!/bin/bash
# complete path under the PMC passes once more, final build (512 streams per launch group: AV_DK_WG=64 forces the large-batch shapes)
set -o pipefail
O=gpurun_out/r05q; mkdir -p $O
AV_DK_WG=64 bash profiles/r05/collect_pmc.sh $O/pmc_final --streams 512 --steps 4 --warmup 2 --no-regimes > $O/pmc_final_stdout.txt 2>&1 || { tail -20 $O/pmc_final_stdout.txt; exit 1; }
python3 - $O/pmc_final/pmc_summary.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
lk=d['lk_track_g16_kernel<15>']; nlk=7.0
steps=lk['dispatches']/nlk
rows=[]
for k,e in d.items():
    if k.startswith('Cijk') or 'rocblas' in k: continue
    per=e['dispatches']/steps
    wc=e.get('SQ_WAVE_CYCLES',0)*per; vg=max(e.get('vgpr',0),1)
    rows.append((wc*vg, k, per, wc, vg, e.get('lds',0), e.get('SQ_INSTS_VALU',0)*per, (e.get('FETCH_SIZE',0)*2+e.get('WRITE_SIZE',0))*per/1e3))
tot=sum(r[0] for r in rows); toti=sum(r[6] for r in rows); totb=sum(r[7] for r in rows)
print('steps in the run: %.1f' % steps)
print('%-30s %6s %11s %5s %6s %9s %8s %8s' % ('kernel','n/step','Mwavecyc/st','vgpr','lds','occ-share','valu-sh','hbm-sh'))
for v,k,per,wc,vg,lds,vi,mb in sorted(rows,reverse=True):
    print('%-30s %6.1f %11.1f %5d %6d %9.3f %8.3f %8.3f' % (k[:30],per,wc/1e6,vg,lds,v/tot,vi/toti,mb/totb))
PY
