#!/bin/bash
# LDS bank conflicts of the front-end's kernels (one SQ pass)
set -o pipefail
ROOT=$GRAFT_REPO_ROOT; O=$ROOT/gpurun_out/r05p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmc_lds
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d /tmp/pmc_lds -o p -- python3 $ROOT/bench.py --frontend-only --streams 64 --steps 6 --warmup 2 --no-cpu-baseline --no-regimes > $O/b.json 2> $O/b.err || { tail -20 $O/b.err; exit 1; }
python3 - $(find /tmp/pmc_lds -name "*counter_collection.csv" | head -1) <<'PY'
import csv,sys,re,collections
per=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
seen=set()
for r in csv.DictReader(open(sys.argv[1])):
    k=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name']); k=re.sub(r'\(.*$','',k).replace('void ','').strip()
    if k.startswith('at::') or 'rocclr' in k or 'Cijk' in k or 'elementwise' in k: continue
    per[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if (k,r['Dispatch_Id']) not in seen: seen.add((k,r['Dispatch_Id'])); n[k]+=1
for k,c in sorted(per.items(), key=lambda kv:-kv[1].get('SQ_WAVE_CYCLES',0)):
    d=n[k]; print('%-30s n %4d  LDS insts/wave %7.1f  idx_active %10.0f  bank_conflict %10.0f (%.2f of active)  wait_inst_lds/wave_cycles %.3f  active_inst_lds/wave_cycles %.3f' % (k[:30], d, c['SQ_INSTS_LDS']/max(c['SQ_WAVES'],1), c['SQ_LDS_IDX_ACTIVE']/d, c['SQ_LDS_BANK_CONFLICT']/d, c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_LDS_IDX_ACTIVE'],1), c['SQ_WAIT_INST_LDS']/max(c['SQ_WAVE_CYCLES'],1), c['SQ_ACTIVE_INST_LDS']/max(c['SQ_WAVE_CYCLES'],1)))
PY
