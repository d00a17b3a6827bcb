#!/bin/bash
# A/B builds of gemm.hip timed by rocprofv3 on tools/gg_probe.py cases
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; OUT=gpurun_out/gg_ab; mkdir -p $R/$OUT
export VARIANT_FILE=gemm
bash $R/tools/build_variant.sh w8 > /dev/null
bash $R/tools/build_variant.sh w4 -DJMAC_GG_WAVES=4 -DJMAC_GG_KC=152 > /dev/null
cd /tmp && export TMPDIR=/tmp
for c in "NN 962x300x300" "NN x4" "TN 300x600x962"; do
for v in w8 w4; do
  export JMAC_LIB_PATH=/tmp/jmac_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT -o p -- python3 $R/tools/gg_probe.py "$c" > /dev/null 2>&1
  python3 - "$v $c" <<PY
import csv, sys
for r in csv.DictReader(open("$R/$OUT/p_kernel_stats.csv")):
    if "grouped" in r["Name"]:
        print("%-26s calls %4s avg %7.2f us  min %7.2f" % (sys.argv[1], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done; done
