#!/bin/bash
# ON THE GPU BOX, from the repo root:  bash profiles/r03/run_fetch_calib.sh   -> profiles/r03/fetch_calib.json (via gpurun_out/)
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/fetch_calib; mkdir -p $OUT
[ -x $ROOT/profiles/r03/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $ROOT/profiles/r03/fetch_calib $ROOT/profiles/r03/fetch_calib.hip
cd /tmp && export TMPDIR=/tmp
$ROOT/profiles/r03/fetch_calib 4 > $OUT/plain.jsonl
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/fc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/fc_$c -o p -- $ROOT/profiles/r03/fetch_calib 4 > /tmp/fc_$c.out 2> /tmp/fc_$c.err
  cp $(find /tmp/fc_$c -name "*counter_collection.csv" | head -1) $OUT/${c}_counter_collection.csv
done
python3 - $OUT <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
known = {}
for line in open(out + '/plain.jsonl'):
    d = json.loads(line)
    if 'kernel' in d: known[d['kernel']] = d
    else: meta = d
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open('%s/%s_counter_collection.csv' % (out, c))):
        k = row['Kernel_Name'].split('(')[0]
        per[k][row['Dispatch_Id']] += float(row['Counter_Value'])
    for k, disp in per.items():
        res.setdefault(k, {})[c + '_KB_mean_per_dispatch'] = sum(disp.values()) / len(disp)
summary = {'meta': meta, 'kernels': {}}
for k, d in known.items():
    kk = 'sparse' if k.startswith('sparse') else k
    e = dict(d)
    summary['kernels'][k] = e
# sparse128 / sparse64 are the same kernel symbol: dispatch order separates them
per = collections.defaultdict(list)
for row in csv.DictReader(open(out + '/FETCH_SIZE_counter_collection.csv')):
    per[(row['Kernel_Name'].split('(')[0], int(row['Dispatch_Id']))].append(float(row['Counter_Value']))
order = sorted(per)
byk = collections.defaultdict(list)
for (k, disp) in sorted(per, key=lambda t: t[1]):
    byk[k].append(sum(per[(k, disp)]))
names = {'read16': ['read16'], 'read4': ['read4'], 'read1': ['read1'], 'sparse': ['sparse128', 'sparse64'], 'rows32': ['rows32']}
for sym, vals in byk.items():
    if sym not in names: continue
    labels = names[sym]; n = len(vals) // len(labels)
    for i, lab in enumerate(labels):
        v = vals[i * n:(i + 1) * n]
        e = summary['kernels'][lab]
        e['FETCH_SIZE_bytes_raw'] = sum(v) / len(v) * 1024
        e['raw_over_known'] = e['FETCH_SIZE_bytes_raw'] / e['known_bytes']
lines = meta['buffer_bytes'] / 128
summary['kernels']['sparse128']['raw_bytes_per_128B_line'] = summary['kernels']['sparse128']['FETCH_SIZE_bytes_raw'] / lines
summary['kernels']['sparse64']['raw_bytes_per_64B_half_line'] = summary['kernels']['sparse64']['FETCH_SIZE_bytes_raw'] / (2 * lines)
summary['kernels']['rows32']['unique_bytes_upper_bound'] = meta['images'] * 768 * 512
summary['kernels']['rows32']['raw_over_unique'] = summary['kernels']['rows32']['FETCH_SIZE_bytes_raw'] / (meta['images'] * 768 * 512)
json.dump(summary, open(out + '/fetch_calib.json', 'w'), indent=1)
print(json.dumps(summary, indent=1))
PY
