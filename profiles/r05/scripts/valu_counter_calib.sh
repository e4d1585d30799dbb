#!/bin/bash
# what does SQ_INSTS_VALU count?  The microbench's kernels execute a KNOWN number of VALU instructions per wave (n x 32 + a prologue)
set -o pipefail
ROOT=$GRAFT_REPO_ROOT; O=$ROOT/gpurun_out/r05r; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmc_cal
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_cal -o p -- $ROOT/profiles/r05/valu_issue_microbench > $O/mb.json 2> $O/mb.err || { tail -20 $O/mb.err; exit 1; }
python3 - $(find /tmp/pmc_cal -name "*counter_collection.csv" | head -1) > $O/valu_counter_calib.txt <<'PY'
import csv,sys,re,collections
per=collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    per[(r['Kernel_Name'], r['Dispatch_Id'])][r['Counter_Name']] = per[(r['Kernel_Name'], r['Dispatch_Id'])].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    per[(r['Kernel_Name'], r['Dispatch_Id'])]['wg'] = int(r['Workgroup_Size'])
rows=collections.defaultdict(list)
for (k,d),c in per.items():
    if c.get('SQ_WAVES',0) > 0: rows[k].append((int(d), c))
print('kernel: workgroup size, SQ_INSTS_VALU per wave (expected 64,000 + prologue for the n = 2,000 dispatches), SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU')
for k,v in rows.items():
    v.sort()
    name=re.sub(r'\(.*$','',k).replace('k_','')
    out=[]
    for d,c in v:
        ipw=c['SQ_INSTS_VALU']/c['SQ_WAVES']
        if ipw > 10000: out.append('%d: %.0f (active/inst %.2f)' % (c['wg'], ipw, c.get('SQ_ACTIVE_INST_VALU',0)/max(c['SQ_INSTS_VALU'],1)))
    print('%-26s %s' % (name[:26], '  '.join(out)))
PY
head -30 $O/valu_counter_calib.txt
