#!/bin/bash
# rocprofv3 PMC passes for the aggregation kernels, one counter group per pass (MI355X_MICROARCH.md: FETCH_SIZE and
# WRITE_SIZE do not fit one pass; --pmc runs carry no trace options).  usage (repo root, GPU box):
#   bash tools/closed/pmc_collect_r3.sh <out_dir>      then   python3 tools/closed/pmc_summarize_r2.py <out_dir> > <out_dir>/summary.json
OUT=${1:-gpurun_out/pmc_r3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name counters... -- program args
  name=$1; shift; ctr=$1; shift
  timeout 600 rocprofv3 --pmc $ctr --output-format csv -d $R/$OUT/$name -o p -- "$@" > $R/$OUT/$name.log 2>&1
}
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '+')
  run c4_$tag "$c" python3 $R/tools/agg_sweep.py 1.0 auto 300 1
  DBG=samerel run c4samerel_$tag "$c" python3 $R/tools/agg_sweep.py 1.0 auto 300 0
  run ja_$tag "$c" python3 $R/tools/ja_sweep.py ja-real
done
cd $R
python3 tools/closed/pmc_summarize_r2.py $OUT > $OUT/summary.json
find $OUT -name "*.csv" -size +2M -delete
head -c 600 $OUT/summary.json
