#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/r05j; mkdir -p $O
for t in 1024 512 256 1024 512; do
AV_PYR_L23_T=$t python bench.py --frontend-only --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_fe_l23_${t}_$RANDOM.json 2> $O/e.txt; echo "fe l23=$t rc $?"
done
for f in $O/bench_fe_l23_*.json; do python3 - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], 'value %.0f ms/step %.3f kernels %s' % (d['value'], d['ms_per_step'], json.dumps(d.get('kernel_ms_per_step'))))
PY
done
