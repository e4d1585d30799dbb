#!/bin/bash
# forward + backward stereo LK in one launch (AV_LK_PAIR), executed-flop counters, pinned-occupancy microbench
set -o pipefail
O=$PWD/gpurun_out/r05h; mkdir -p $O
./profiles/r05/valu_issue_microbench > $O/valu_issue_microbench.json 2> $O/mb.err || { tail -5 $O/mb.err; exit 1; }
python3 - $O/valu_issue_microbench.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for r in d['rows']: print("%-28s w1 %6.2f  w2 %6.2f  w4 %6.2f  w5 %6.2f  ticks/us %s" % (r["op"], r["w1"], r["w2"], r["w4"], r["w5"], r.get("ticks_per_us")))
PY
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_frontend.py tests/test_gpu_pipeline.py tests/test_gpu_bench.py -m gpu -x -q > $O/pytest_subset.txt 2>&1; rc=$?; tail -4 $O/pytest_subset.txt; [ $rc -eq 0 ] || exit 1
for pr in 1 0 1 0; do
AV_LK_PAIR=$pr python bench.py --frontend-only --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_fe_pair${pr}_$RANDOM.json 2> $O/e.txt; echo "fe pair=$pr rc $?"
done
for pr in 1 0; do
AV_LK_PAIR=$pr python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_pair$pr.json 2> $O/e_pair$pr.txt; echo "full pair=$pr rc $?"; tail -2 $O/e_pair$pr.txt
done
for f in $O/bench_fe_pair*.json $O/bench_pair1.json $O/bench_pair0.json; do python3 - $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline']
    print(sys.argv[1].split('/')[-1], 'value %.0f ms/step %.2f fe_only %s lk_ms %.3f x%d (alone %s) frac %.4f kernels %s' % (d['value'], d['ms_per_step'], d.get('frontend_only_frames_per_s'), r['avg_launch_ms'], r['launches'], r.get('avg_launch_ms_frontend_only'), r['frac'], json.dumps(d.get('kernel_ms_per_step'))))
    m=d.get('roofline_msckf')
    if m: print('   msckf frac %.4f executed %s' % (m['frac'], json.dumps(m.get('executed'))))
    if d.get('regimes'): print('   regimes', json.dumps(d['regimes'])[:900])
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY
done
