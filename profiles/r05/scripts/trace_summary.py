import csv, sys, re, collections, statistics as st
path = sys.argv[1]; last_frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rows = []
for r in csv.DictReader(open(path)):
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); n = re.sub(r'\(.*$', '', n).replace('void ', '')
    if n.startswith('at::') or 'rocblas' in n or n.startswith('Cijk') or 'elementwise' in n: continue
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n, r.get('Stream_Id', r.get('Queue_Id', '?'))))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
cut = t1 - (t1 - t0) * last_frac
sel = [r for r in rows if r[0] >= cut]
by = collections.defaultdict(list)
for a, b, n, q in sel: by[n].append((b - a) / 1e3)
print('window %.1f ms, %d dispatches' % ((t1 - cut) / 1e6, len(sel)))
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print('%-34s n %5d  sum %9.2f ms  median %8.1f us  p90 %8.1f  max %8.1f' % (n[:34], len(v), sum(v) / 1e3, st.median(v), sorted(v)[int(0.9 * (len(v) - 1))], max(v)))
if len(sys.argv) > 3:     # timeline of one queue: consecutive dispatches with gaps
    q = sys.argv[3]
    seq = [r for r in sel if r[3] == q][:120]
    prev = None
    for a, b, n, _ in seq:
        print('%-28s dur %8.1f us  gap %8.1f us' % (n[:28], (b - a) / 1e3, (a - prev) / 1e3 if prev else 0)); prev = b
else:
    qs = collections.Counter(r[3] for r in sel); print('queues:', qs.most_common(12))
