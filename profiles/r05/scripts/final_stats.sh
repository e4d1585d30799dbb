#!/bin/bash
# rocprofv3 kernel stats of the driver command (re-run on its own: the run inside final_b.sh hit a 30 ms stall)
set -o pipefail
O=$PWD/gpurun_out/r05_final; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pf
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_driver_cmd_s2048_under_rocprof_r05.json 2> $O/e6.txt || { tail -5 $O/e6.txt; exit 1; }
python3 - $(find /tmp/pf -name "*kernel_stats.csv" | head -1) $O/bench_driver_cmd_s2048_r05_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
keep=[r for r in rows if not r['Name'].startswith('void at::') and 'at::native' not in r['Name'] and 'elementwise' not in r['Name'] and 'rocblas' not in r['Name'] and not r['Name'].startswith('Cijk')]
w=csv.DictWriter(open(sys.argv[2],'w'),fieldnames=rows[0].keys()); w.writeheader()
for r in keep: w.writerow(r)
for r in keep:
    if 'lk_track' in r['Name']: print('LK', r['Calls'], r['AverageNs'], r.get('MaxNs'))
PY
python3 $R/profiles/r05/scripts/trace_summary.py $(find /tmp/pf -name "*kernel_trace.csv" | head -1) 1.0 > $O/driver_cmd_kernel_medians_r05.txt; head -8 $O/driver_cmd_kernel_medians_r05.txt
python3 -c "
import json; d=json.load(open('$O/bench_driver_cmd_s2048_under_rocprof_r05.json')); print('under rocprof: value %.0f lk events %.3f ms' % (d['value'], d['roofline']['avg_launch_ms']))"
