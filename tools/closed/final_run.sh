#!/bin/bash
# One GPU-box pass that regenerates what profiles/ holds for a round: the -m gpu suite, the default bench line, the same
# command under rocprofv3 --kernel-trace --stats, and a kernel trace of the step alone for tools/step_breakdown.py.
#   usage (repo root, GPU box):  bash tools/closed/final_run.sh gpurun_out/<dir>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; OUT=${1:-gpurun_out/final}
mkdir -p $R/$OUT; cd $R
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5) > $OUT/gpu_tests.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o p -- python3 $R/bench.py > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/prof.err
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/prof2 -o p -- python3 $R/bench.py --no-synth --no-cpu-baseline > /dev/null 2> $R/$OUT/prof2.err
cd $R
python3 tools/step_breakdown.py $OUT/prof2/p_kernel_trace.csv 70 > $OUT/step_breakdown.txt 2>&1
rm -f $OUT/prof2/p_kernel_trace.csv $OUT/prof/p_kernel_trace.csv     # tens of MB; the stats CSV and the breakdown are kept
tail -3 $OUT/gpu_tests.log; tail -c 600 $OUT/bench.json; head -3 $OUT/step_breakdown.txt
