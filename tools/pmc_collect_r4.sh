#!/bin/bash
# rocprofv3 PMC passes for the aggregation kernels, one counter group per pass (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do
# not fit one pass; --pmc runs carry no trace options).  Workloads: config 4 fp32 (forward + backward), config 4 bf16 forward
# (padded halves: the half-wave kernel), the same with every edge on relation 0 (calibration of the FETCH_SIZE x2 correction on the
# kernels' own access pattern), the REAL 5-KG union fp32 / bf16, the real ja graph.
#   usage (repo root, GPU box):  bash tools/pmc_collect_r4.sh <out_dir>
OUT=${1:-gpurun_out/pmc_r4}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name counters -- program args
  name=$1; shift; ctr=$1; shift
  timeout 900 rocprofv3 --pmc $ctr --output-format csv -d $R/$OUT/$name -o p -- "$@" > $R/$OUT/$name.log 2>&1
}
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '+')
  run c4_$tag "$c" python3 $R/tools/agg_sweep.py 1.0 auto 300 1
  BF16=1 run c4bf16_$tag "$c" python3 $R/tools/agg_sweep.py 1.0 auto 300 0
  DBG=samerel run c4samerel_$tag "$c" python3 $R/tools/agg_sweep.py 1.0 auto 300 0
  DBG=samerel BF16=1 run c4bf16samerel_$tag "$c" python3 $R/tools/agg_sweep.py 1.0 auto 300 0
  run union_$tag "$c" python3 $R/tools/union_agg_probe.py
  run ja_$tag "$c" python3 $R/tools/ja_sweep.py ja-real
done
cd $R
python3 tools/pmc_summarize_r4.py $OUT > $OUT/summary.json
find $OUT -name "*.csv" -size +2M -delete
head -c 1500 $OUT/summary.json
