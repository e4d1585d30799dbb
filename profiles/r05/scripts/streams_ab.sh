#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/r05l; mkdir -p $O
for S in 3072 2560 2048; do
python bench.py --streams $S --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_s$S.json 2> $O/e_s$S.txt; echo "S=$S rc $?"; tail -2 $O/e_s$S.txt
python3 - $O/bench_s$S.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); m=d['roofline_msckf']
    print(sys.argv[1].split('/')[-1], 'value %.0f ms/step %.2f fe_only %.0f chain %.2f' % (d['value'], d['ms_per_step'], d.get('frontend_only_frames_per_s') or 0, m['chain_ms_per_step']))
except Exception as e: print('unreadable', e)
PY
done
