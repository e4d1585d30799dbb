#!/bin/bash
# Run ON THE GPU BOX:  bash profiles/r05/collect_pmc.sh <out_dir relative to the repo> [bench args...]
# Three separate counter passes (SQ block; FETCH_SIZE; WRITE_SIZE -- TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2:
# MI355X_MICROARCH.md "rocprofv3 PMC slots"), --kernel-trace only, never combined with sys / runtime tracing; `python3` follows
# `--` directly (no launcher hop under the profiler).  --no-cpu-baseline is always appended: CPU-baseline children under the
# profiler would skew nothing but cost minutes.
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
: "${1:?usage: collect_pmc.sh <out_dir> [bench args]}"
OUT=$1; shift
ARGS=${@:---frontend-only --streams 64 --steps 6 --warmup 2}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
    name=$1; shift
    rm -rf /tmp/pmc_$name
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- python3 $ROOT/bench.py $ARGS --no-cpu-baseline > /tmp/pmc_$name.json 2> /tmp/pmc_$name.err \
        || { echo "rocprofv3 pass $name failed:"; tail -20 /tmp/pmc_$name.err; exit 1; }
    cp $(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1) "$ROOT/$OUT/pmc_${name}_counter_collection.csv"
}
run sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $ROOT && python3 profiles/r05/summarize_pmc.py "$OUT"
