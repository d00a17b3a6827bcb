#!/bin/bash
# rocprofv3 --kernel-trace --stats of a python script: tools/closed/kstats.sh <out dir under gpurun_out> <script> [args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; OUT=gpurun_out/$1; shift; mkdir -p $R/$OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT -o p -- python3 $R/"$@" > /dev/null 2> $R/$OUT/err.log
cd $R; python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/p_kernel_stats.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:14]:
    print("%-70s calls %5s avg %9.2f us  total %9.1f us" % (r["Name"].replace("(anonymous namespace)::","")[:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3))
PY
rm -f $OUT/p_kernel_trace.csv
