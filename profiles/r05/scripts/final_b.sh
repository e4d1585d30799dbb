#!/bin/bash
# round-5 final profiles: rocprofv3 kernel stats of the driver command, exclusive and contended timelines of one step, sweep throughput
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
print('kernel stats rows', len(keep))
PY
python3 $R/profiles/r05/scripts/trace_summary.py $(find /tmp/pf -name "*kernel_trace.csv" | head -1) 1.0 > $O/driver_cmd_kernel_medians_r05.txt; head -14 $O/driver_cmd_kernel_medians_r05.txt
cd $R
bash profiles/r05/scripts/exclusive_timeline.sh > $O/filter_step_exclusive_timeline_r05.txt 2>&1 || { tail -5 $O/filter_step_exclusive_timeline_r05.txt; exit 1; }
tail -46 $O/filter_step_exclusive_timeline_r05.txt | head -24
bash profiles/r05/scripts/contended_timeline.sh > /dev/null 2>&1; cp gpurun_out/r05y/contended_timeline.txt $O/filter_step_contended_timeline_r05.txt; head -3 $O/filter_step_contended_timeline_r05.txt
timeout -k 10 400 python profiles/r05/sweep_throughput.py > $O/sweep_throughput_s64_r05.json 2> $O/e7.txt || tail -5 $O/e7.txt; head -c 900 $O/sweep_throughput_s64_r05.json
