#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash profiles/r02/collect_pmc.sh <out_dir> [bench args...]
# Three separate counter passes (SQ block; FETCH_SIZE; WRITE_SIZE -- TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2:
# MI355X_MICROARCH.md "rocprofv3 PMC slots"), kernel-trace only, never combined with sys/runtime tracing.
set -e -o pipefail
OUT=$1; shift
ARGS=${@:---frontend-only --streams 64 --steps 6 --warmup 2 --no-cpu-baseline}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
    name=$1; shift
    rm -rf /tmp/pmc_$name
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > /tmp/pmc_$name.json 2> /tmp/pmc_$name.err
    cp $(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1) "$GRAFT_REPO_ROOT/$OUT/pmc_${name}_counter_collection.csv"
}
run sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $GRAFT_REPO_ROOT && python3 profiles/r02/summarize_pmc.py "$OUT"
