#!/bin/bash
# round-5 opening measurement: gpu tests, driver-command bench, exclusive filter timeline
set -o pipefail
O=$PWD/gpurun_out/r05a; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd_s2048.json 2> $O/e1.txt || { tail -5 $O/e1.txt; exit 1; }
python3 - $O/bench_driver_cmd_s2048.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print('value %.0f ms/step %.2f fe_only %s lk_ms %.3f frac %.4f msckf %s' % (d['value'], d['ms_per_step'], d.get('frontend_only_frames_per_s'), r['avg_launch_ms'], r['frac'], json.dumps(d.get('roofline_msckf'))[:600]))
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/t3
AV_BENCH_SERIAL=1 AV_MSCKF_GROUPS=1 timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/t3 -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/serial_g1.json 2> $O/serial_g1.err || { tail -20 $O/serial_g1.err; exit 1; }
python3 $R/profiles/r04/scripts/trace_summary.py $(find /tmp/t3 -name "*kernel_trace.csv" | head -1) 1.0 > $O/serial_g1_summary.txt; head -60 $O/serial_g1_summary.txt
