#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/r05x; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/t3
AV_BENCH_SERIAL=1 AV_MSCKF_GROUPS=1 timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/t3 -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/serial_g1.json 2> $O/serial_g1.err || { tail -20 $O/serial_g1.err; exit 1; }
python3 $R/profiles/r05/scripts/trace_summary.py $(find /tmp/t3 -name "*kernel_trace.csv" | head -1) 1.0 > $O/serial_g1_summary.txt; head -45 $O/serial_g1_summary.txt
python3 - $(find /tmp/t3 -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, re, sys, collections
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    n=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name']); n=re.sub(r'\(.*$','',n).replace('void ','')
    if n.startswith('at::') or 'rocblas' in n or n.startswith('Cijk') or 'elementwise' in n: continue
    rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),n))
rows.sort()
idx=[i for i,r in enumerate(rows) if r[2].startswith("dk_begin")]
i0=idx[40] if len(idx)>41 else idx[0]; i1=idx[41] if len(idx)>41 else len(rows)-1
prev=None; t0=rows[i0][0]
for a,b,n in rows[i0:i1]:
    print('%-26s start %8.1f us dur %8.1f gap %8.1f' % (n[:26], (a-t0)/1e3, (b-a)/1e3, (a-prev)/1e3 if prev else 0)); prev=max(prev or 0,b)
PY
