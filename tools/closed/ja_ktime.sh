#!/bin/bash
# kernel-level durations (rocprofv3 --kernel-trace --stats) of the aggregation kernels on the ja shape
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; OUT=${1:-gpurun_out/ja_ktime}; shift
mkdir -p $R/$OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT -o p -- python3 $R/tools/ja_sweep.py ja "$@" > /dev/null 2>&1
cd $R; python3 - <<PY
import csv
tot=0
for r in csv.DictReader(open("$OUT/p_kernel_stats.csv")):
    n=r["Name"]
    if any(k in n for k in ("rel_attn","sum_parts","reduce_rows","colsum","finalize")):
        a=float(r["AverageNs"])/1e3
        if "fwd" not in n: tot+=a
        print("%-56s calls %4s avg %7.2f us" % (n.replace("void (anonymous namespace)::","").replace("(anonymous namespace)::","")[:56], r["Calls"], a))
print("backward kernels total %.1f us" % tot)
PY
