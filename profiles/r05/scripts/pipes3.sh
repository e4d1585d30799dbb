#!/bin/bash
O=$PWD/gpurun_out/r05q; mkdir -p $O
for a in "--pipelines 1" "--pipelines 2" "--pipelines 4"; do
python bench.py --frontend-only --no-cpu-baseline --steps 20 --warmup 5 $a > $O/bench_v.json 2> $O/e1.txt || { tail -8 $O/e1.txt; exit 1; }
python3 - $O/bench_v.json "$a" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print('fe-only %-14s value %.0f ms/step %.2f kernels %s' % (sys.argv[2], d['value'], d['ms_per_step'], json.dumps({k: round(v,2) for k,v in d.get('kernel_ms_per_step',{}).items()})))
PY
done
