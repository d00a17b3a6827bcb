#!/bin/bash
# A/B builds of gemm.hip (debug macros) timed by rocprofv3 on one case of tools/gg_probe.py
R=$GRAFT_REPO_ROOT; OUT=gpurun_out/gg_ab; mkdir -p $R/$OUT
export VARIANT_FILE=gemm
bash $R/tools/build_variant.sh base > /dev/null
bash $R/tools/build_variant.sh noepi -DGG_NO_EPI > /dev/null
bash $R/tools/build_variant.sh nobody -DGG_NO_BODY > /dev/null
bash $R/tools/build_variant.sh nothing -DGG_NO_BODY -DGG_NO_EPI > /dev/null
cd /tmp && export TMPDIR=/tmp
for v in base noepi nobody nothing; do
  export JMAC_LIB_PATH=/tmp/jmac_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT -o p -- python3 $R/tools/gg_probe.py "${1:-NN 962x300x300}" > /dev/null 2>&1
  python3 - $v <<PY
import csv, sys
for r in csv.DictReader(open("$R/$OUT/p_kernel_stats.csv")):
    if "grouped" in r["Name"]:
        print("%-10s calls %4s avg %7.2f us  min %7.2f" % (sys.argv[1], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done
