#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export VARIANT_FILE=gemm
bash $R/tools/build_variant.sh trace -DJMAC_GG_TRACE > /dev/null
export JMAC_LIB_PATH=/tmp/jmac_trace.so
for n in 1 4; do echo "== $n task(s)"; python3 $R/tools/closed/gg_trace.py $n 2>&1 | grep "phase\|wall"; done
for n in 1 2; do echo "== $n TN task(s)"; python3 $R/tools/closed/gg_trace.py $n tn 2>&1 | grep "phase\|wall"; done
