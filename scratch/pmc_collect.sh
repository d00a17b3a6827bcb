#!/bin/bash
# rocprofv3 PMC passes (one counter per pass, as MI355X_MICROARCH.md prescribes) for the aggregation kernels.
# usage: scratch/pmc_collect.sh <out_dir>      (run from the repo root on the GPU box)
OUT=${1:-gpurun_out/pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/$OUT/ja_$c -o p -- python3 $R/scratch/ja_sweep.py ja > $R/$OUT/ja_$c.log 2>&1
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $R/$OUT/c4_$c -o p -- python3 $R/scratch/agg_sweep.py 1.0 512 300 1 > $R/$OUT/c4_$c.log 2>&1
  BF16=1 timeout 600 rocprofv3 --pmc $c --output-format csv -d $R/$OUT/c4bf16_$c -o p -- python3 $R/scratch/agg_sweep.py 1.0 512 300 0 > $R/$OUT/c4bf16_$c.log 2>&1
done
ls -R $R/$OUT | head -40
