#!/bin/bash
O=$PWD/gpurun_out/r05p; mkdir -p $O
timeout -k 10 120 ./profiles/r05/mfma_f64_probe > $O/mfma_f64_probe.json 2> $O/probe.err || { tail -5 $O/probe.err; exit 1; }
cat $O/mfma_f64_probe.json
