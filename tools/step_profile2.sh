#!/bin/bash
# Kernel trace of one workload's hipGraph-replayed step (tools/pair_probe.py: --batched 1|0 = the el+ja pair step, --ja = the headline
# single-KG step) -> tools/step_breakdown.py.   usage (repo root, GPU box): bash tools/step_profile2.sh gpurun_out/<dir> <probe args>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; OUT=${1:-gpurun_out/step2}; shift
mkdir -p $R/$OUT; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/prof -o p -- python3 $R/tools/pair_probe.py "$@" > $R/$OUT/probe.json 2> $R/$OUT/prof.err
cd $R
python3 tools/step_breakdown.py $OUT/prof/p_kernel_trace.csv 70 ${STEP_GRID:+grid} > $OUT/step_breakdown.txt 2>&1
rm -f $OUT/prof/p_kernel_trace.csv
head -3 $OUT/step_breakdown.txt; cat $OUT/probe.json
