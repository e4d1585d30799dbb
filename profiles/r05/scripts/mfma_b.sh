#!/bin/bash
# complete path under the three PMC passes (512 streams: the 2,048-stream run segfaults inside rocprofv3 on a torch copy) + the contended timeline
set -o pipefail
O=gpurun_out/r05q; mkdir -p $O
bash profiles/r05/collect_pmc.sh $O/pmc_full --streams 512 --steps 4 --warmup 2 --no-regimes > $O/pmc_full_stdout.txt 2>&1 || { tail -20 $O/pmc_full_stdout.txt; exit 1; }
python3 - $O/pmc_full/pmc_summary.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); steps=6.0
rows=[]
for k,e in d.items():
    if k.startswith('Cijk') or 'rocblas' in k: continue
    per=e['dispatches']/steps
    rows.append((e.get('SQ_INSTS_VALU',0)*per, k, per, e))
tot=sum(r[0] for r in rows)
print('%-28s %6s %10s %6s %9s %9s %8s %6s %5s' % ('kernel','n/step','Minst/step','share','fetchMB*2','writeMB','wait_any','vgpr','lds'))
for v,k,per,e in sorted(rows,reverse=True):
    print('%-28s %6.1f %10.1f %6.3f %9.1f %9.1f %8.2f %6d %5d' % (k[:28],per,v/1e6,v/tot,e.get('FETCH_SIZE',0)*per*2/1e3,e.get('WRITE_SIZE',0)*per/1e3,e.get('wait_any_share',0),e.get('vgpr',0),e.get('lds',0)))
print('total Minst/step %.1f' % (tot/1e6))
PY
bash profiles/r05/scripts/contended_timeline.sh && cat gpurun_out/r05y/contended_timeline.txt
