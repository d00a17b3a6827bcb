#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
for H in 0 1; do JMAC_FWD_HOT=$H python tools/closed/c4_probe.py 2>/dev/null; done
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; OUT=gpurun_out/pmc_r4_hot; mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '+')
  JMAC_FWD_HOT=1 timeout 900 rocprofv3 --pmc $c --output-format csv -d $R/$OUT/c4_$tag -o p -- python3 $R/tools/agg_sweep.py 1.0 auto 300 0 > $R/$OUT/c4_$tag.log 2>&1
  JMAC_FWD_HOT=1 BF16=1 timeout 900 rocprofv3 --pmc $c --output-format csv -d $R/$OUT/c4bf16_$tag -o p -- python3 $R/tools/agg_sweep.py 1.0 auto 300 0 > $R/$OUT/c4bf16_$tag.log 2>&1
done
cd $R
python3 tools/pmc_summarize_r4.py $OUT > $OUT/summary.json
find $OUT -name "*.csv" -size +2M -delete
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/pmc_r4_hot/summary.json"))
for k in ("c4","c4bf16"):
    for kn,kv in d[k].items():
        print(k, kn[:60], "traffic %.2f GB" % (kv["traffic_bytes_corrected"]/1e9), "L2 hit", kv["l2_hit_rate"])
PY
