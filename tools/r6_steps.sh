#!/bin/bash
# round 6: kernel traces of the three replayed training steps (real ja single-KG step = the headline, real el + ja pair step,
# five-KG forward_stacked step) -> gpurun_out/r6_step_kernels.json (machine-readable: bench.py's roofline.frac_in_step reads the
# committed copy, profiles/r6_step_kernels.json) + gpurun_out/r6_step_breakdown.txt.   usage (repo root, GPU box): bash tools/r6_steps.sh
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; mkdir -p gpurun_out; rm -f gpurun_out/r6_step_kernels.json
: > gpurun_out/r6_step_breakdown.txt
for spec in "ja:--ja" "pair:--batched 1" "union_train:--union"; do
  key=${spec%%:*}; args=${spec#*:}
  OUT=gpurun_out/r6_step_$key; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/prof -o p -- python3 $R/tools/pair_probe.py $args > $R/$OUT/probe.json 2> $R/$OUT/prof.err )
  python3 tools/step_breakdown.py $OUT/prof/p_kernel_trace.csv 70 --json gpurun_out/r6_step_kernels.json $key > $OUT/step_breakdown.txt 2>&1
  rm -f $OUT/prof/p_kernel_trace.csv
  { echo "== $key step (hipGraph replay; tools/pair_probe.py $args)"; cat $OUT/step_breakdown.txt; cat $OUT/probe.json; echo; } >> gpurun_out/r6_step_breakdown.txt
done
head -4 gpurun_out/r6_step_breakdown.txt
