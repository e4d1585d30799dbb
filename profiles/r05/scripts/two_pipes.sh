#!/bin/bash
# two independent 1,024-stream pipelines (own engine, own filter) on one GPU, free-running: does de-phasing the two halves pay?
set -o pipefail
O=$PWD/gpurun_out/r05q; mkdir -p $O
python bench.py --streams 1024 --steps 40 --warmup 10 --no-cpu-baseline --no-regimes > $O/two_a.json 2> $O/two_a.err &
PA=$!
python bench.py --streams 1024 --steps 40 --warmup 10 --no-cpu-baseline --no-regimes > $O/two_b.json 2> $O/two_b.err &
PB=$!
wait $PA; RA=$?; wait $PB; RB=$?
[ $RA -eq 0 ] && [ $RB -eq 0 ] || { tail -5 $O/two_a.err $O/two_b.err; exit 1; }
python3 - $O/two_a.json $O/two_b.json <<'PY'
import json,sys
a=json.load(open(sys.argv[1])); b=json.load(open(sys.argv[2]))
print('two free-running 1,024-stream pipelines: %.0f + %.0f = %.0f frames/s (ms/step %.2f / %.2f)' % (a['value'], b['value'], a['value']+b['value'], a['ms_per_step'], b['ms_per_step']))
PY
python bench.py --streams 1024 --steps 40 --warmup 10 --no-cpu-baseline --no-regimes > $O/one.json 2> $O/one.err && python3 -c "
import json; d=json.load(open('$O/one.json')); print('one 1,024-stream pipeline alone: %.0f frames/s, %.2f ms/step' % (d['value'], d['ms_per_step']))"
