#!/bin/bash
# launch shapes by batch size: parity of both shapes, then the default line with its small-batch regimes and the configs[4] shape
set -o pipefail
O=$PWD/gpurun_out/r05q; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_msckf.py -x -q -m gpu -k "golden_and_oracle or blank or reset or capacity" > $O/pytest_msckf.txt 2>&1 || { tail -30 $O/pytest_msckf.txt; exit 1; }
tail -2 $O/pytest_msckf.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_o.json 2> $O/e1.txt || { tail -5 $O/e1.txt; exit 1; }
python bench.py --grid 10 15 10 --streams 256 --steps 20 --warmup 5 --no-cpu-baseline --no-regimes > $O/bench_o5.json 2> $O/e2.txt || { tail -5 $O/e2.txt; exit 1; }
python3 - $O/bench_o.json $O/bench_o5.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print('driver %.0f ms/step %.2f regimes %s' % (d['value'], d['ms_per_step'], [(g['name'], round(g['ms_per_step'],3), round(g['stream_frames_per_s'])) for g in d['regimes']]))
d=json.load(open(sys.argv[2])); print('config5 %.0f ms/step %.2f' % (d['value'], d['ms_per_step']))
PY
