#!/bin/bash
O=$PWD/gpurun_out/r05q; mkdir -p $O
timeout -k 10 600 python profiles/r05/scripts/two_engines.py > $O/two_engines.json 2> $O/two_engines.err || { tail -15 $O/two_engines.err; exit 1; }
cat $O/two_engines.json
