#!/bin/bash
O=$PWD/gpurun_out/r05n; mkdir -p $O
./profiles/r05/valu_issue_microbench > $O/valu_issue_microbench.json 2> $O/mb.err || { tail -5 $O/mb.err; exit 1; }
python3 - $O/valu_issue_microbench.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for r in d['rows']: print("%-28s w1 %6.2f  w2 %6.2f  w3 %6.2f  w4 %6.2f  w4_ns %s  launch/loop %s" % (r["op"], r["w1"], r["w2"], r["w3"], r["w4"], r.get("w4_ns"), r.get("launch_over_loop")))
PY
