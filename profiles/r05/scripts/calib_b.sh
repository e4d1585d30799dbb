#!/bin/bash
# clock calibration of the stamps / microbench (s_memtime against the constant 100 MHz s_memrealtime), LK phase stamps, contended timeline
set -o pipefail
O=$PWD/gpurun_out/r05d; mkdir -p $O
./profiles/r05/valu_issue_microbench > $O/valu_issue_microbench.json 2> $O/mb.err || { tail -5 $O/mb.err; exit 1; }
python3 - $O/valu_issue_microbench.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for r in d['rows']: print("%-28s w1 %6.2f  w2 %6.2f  w4 %6.2f  w5 %6.2f  w5 %.3f ns (events) %.3f ns (in-kernel)  ticks/us %s" % (r["op"], r["w1"], r["w2"], r["w4"], r["w5"], r["w5_ns"], r.get("w5_loop_ns"), r.get("ticks_per_us")))
PY
AV_LK_PROF=1 python bench.py --frontend-only --steps 10 --warmup 3 --no-cpu-baseline --no-regimes > $O/bench_fe_prof.json 2> $O/lk_phase_stamps.txt; echo "prof rc $?"; grep AV_LK_PROF $O/lk_phase_stamps.txt
python3 -c "
import json; d=json.load(open('$O/bench_fe_prof.json')); print('profiled front-end-only: value %.0f lk_ms %.3f' % (d['value'], d['roofline']['avg_launch_ms']))"
bash profiles/r05/scripts/contended_timeline.sh
