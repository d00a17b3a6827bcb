#!/bin/bash
# Kernel trace of the headline step alone -> tools/step_breakdown.py (per-step busy time by kernel family).
#   usage (repo root, GPU box):  bash tools/closed/step_profile.sh gpurun_out/<dir> [bench.py args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; OUT=${1:-gpurun_out/step}; shift
mkdir -p $R/$OUT; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/prof -o p -- python3 $R/bench.py --no-synth --no-cpu-baseline "$@" > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/prof.err
cd $R
python3 tools/step_breakdown.py $OUT/prof/p_kernel_trace.csv 80 > $OUT/step_breakdown.txt 2>&1
python3 tools/closed/grouped_in_step.py $OUT/prof/p_kernel_trace.csv > $OUT/grouped_in_step.txt 2>&1
rm -f $OUT/prof/p_kernel_trace.csv
head -4 $OUT/step_breakdown.txt
