#!/bin/bash
# kernel-level durations (rocprofv3 --kernel-trace --stats) of the grouped GEMM by case of tools/gg_probe.py
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; OUT=gpurun_out/gg_ktime; mkdir -p $R/$OUT; cd /tmp && export TMPDIR=/tmp
for c in "NN 962x300x300" "NN x4" "NT 962x300x600" "TN 300x300x962" "TN 300x600x962"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT -o p -- python3 $R/tools/gg_probe.py "$c" > /dev/null 2>&1
  python3 - "$c" <<PY
import csv, sys
for r in csv.DictReader(open("$R/$OUT/p_kernel_stats.csv")):
    if "grouped" in r["Name"]:
        print("%-22s calls %4s avg %7.2f us  min %7.2f  max %7.2f" % (sys.argv[1], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done
