#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/r05y; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/t4
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/t4 -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/b.json 2> $O/b.err || { tail -20 $O/b.err; exit 1; }
python3 - $(find /tmp/t4 -name "*kernel_trace.csv" | head -1) > $O/contended_timeline.txt <<'PY'
import csv, re, sys, collections
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    n=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name']); n=re.sub(r'\(.*$','',n).replace('void ','')
    if n.startswith('at::') or 'rocblas' in n or n.startswith('Cijk') or 'elementwise' in n: continue
    rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),n,r['Queue_Id']))
rows.sort()
isf=lambda n: n.startswith(('dk_','upd_','feature_kernel','triangulate'))
beg=[i for i,r in enumerate(rows) if r[2].startswith('dk_begin')]
# two consecutive group-steps of the same queue in the timed region
i0=beg[len(beg)*3//4]; q=rows[i0][3]
nxt=[i for i in beg if i>i0 and rows[i][3]==q][0]
t0=rows[i0][0]; t1=rows[nxt][0]
print('period of one group (dk_begin to next dk_begin on queue %s): %.2f ms' % (q,(t1-t0)/1e6))
win=[r for r in rows if r[0]>=t0 and r[0]<t1]
# union lengths
def union(iv):
    iv=sorted(iv); tot=0; cs,ce=None,None
    for a,b in iv:
        if cs is None: cs,ce=a,b
        elif a<=ce: ce=max(ce,b)
        else: tot+=ce-cs; cs,ce=a,b
    if cs is not None: tot+=ce-cs
    return tot
fe=[(a,min(b,t1)) for a,b,n,qq in win if not isf(n) and 'rocclr' not in n]
fl=[(a,min(b,t1)) for a,b,n,qq in win if isf(n)]
U=lambda x: union(x)/1e6
print('front-end kernels active %.2f ms, filter kernels active %.2f ms, either %.2f ms, both %.2f ms' % (U(fe),U(fl),U(fe+fl),U(fe)+U(fl)-U(fe+fl)))
print('sum of durations: front-end %.2f ms, filter %.2f ms' % (sum(b-a for a,b in fe)/1e6, sum(b-a for a,b in fl)/1e6))
for a,b,n,qq in win:
    if 'rocclr' in n: continue
    print('%-28s q%-3s start %8.1f us dur %8.1f' % (n[:28], qq, (a-t0)/1e3, (b-a)/1e3))
PY
head -4 $O/contended_timeline.txt
python3 -c "
import json; d=json.load(open('$O/b.json')); print('under rocprof: value %.0f ms/step %.2f' % (d['value'], d['ms_per_step']))"
