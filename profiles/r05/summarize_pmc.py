"""Per-kernel means of the rocprofv3 --pmc passes written by collect_pmc.sh -> <dir>/pmc_summary.json (small, committed;
the raw counter_collection.csv files are dropped).  FETCH_SIZE / WRITE_SIZE are in KB as rocprofv3 reports them, raw
(uncorrected: bench.py applies the x2 of MI355X_MICROARCH.md to FETCH_SIZE, confirmed for this access pattern by fetch_calib.json)."""
import csv
import json
import os
import re
import sys
from collections import defaultdict

d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for name in ('sq', 'fetch', 'write'):
    path = os.path.join(d, 'pmc_%s_counter_collection.csv' % name)
    if not os.path.exists(path):
        continue
    per = defaultdict(lambda: defaultdict(float))        # (kernel, dispatch) -> counter -> value (summed over rows of one dispatch)
    meta = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            k = re.sub(r'\(anonymous namespace\)::', '', row['Kernel_Name'])
            k = re.sub(r'\(.*$', '', k).replace('void ', '').strip()
            key = (k, row['Dispatch_Id'])
            per[key][row['Counter_Name']] += float(row['Counter_Value'])
            meta[k] = dict(vgpr=int(row['VGPR_Count']), lds=int(row['LDS_Block_Size']), wg=int(row['Workgroup_Size']))
    for (k, _disp), cs in per.items():
        for c, v in cs.items():
            acc[k][c].append(v)
    for k, m in meta.items():
        acc[k]['_meta'] = m
out = {}
for k, cs in acc.items():
    if k.startswith('at::') or 'rocclr' in k:
        continue
    e = {c: sum(v) / len(v) for c, v in cs.items() if c != '_meta'}
    e['dispatches'] = max(len(v) for c, v in cs.items() if c != '_meta')
    e.update(cs.get('_meta', {}))
    if e.get('SQ_WAVES'):
        e['valu_per_wave'] = e['SQ_INSTS_VALU'] / e['SQ_WAVES']
        e['wait_any_share'] = e['SQ_WAIT_ANY'] / e['SQ_WAVE_CYCLES']
        e['wait_inst_share'] = e['SQ_WAIT_INST_ANY'] / e['SQ_WAVE_CYCLES']
    out[k] = e
json.dump(out, open(os.path.join(d, 'pmc_summary.json'), 'w'), indent=1, sort_keys=True)
for name in ('sq', 'fetch', 'write'):
    p = os.path.join(d, 'pmc_%s_counter_collection.csv' % name)
    if os.path.exists(p):
        os.remove(p)
print(json.dumps({k: {c: v for c, v in e.items() if c in ('SQ_INSTS_VALU', 'SQ_WAVES', 'valu_per_wave', 'FETCH_SIZE', 'WRITE_SIZE', 'wait_any_share', 'wait_inst_share', 'dispatches')} for k, e in out.items()}, indent=1))
