#!/bin/bash
# round-4 checkpoint on the GPU box: the whole GPU suite, the default bench line, and the kernel-trace stats of the same command
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r4_gpu_tests.log
cat gpurun_out/r4_gpu_tests.log
python bench.py > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err
tail -c 1500 gpurun_out/r4_bench.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_prof_bench -o p -- python3 $R/bench.py > $R/gpurun_out/r4_bench_under_rocprof.json 2> $R/gpurun_out/r4_prof_bench.err
cd $R
find gpurun_out/r4_prof_bench -name "*kernel_trace.csv" -size +4M -delete
head -12 $(find gpurun_out/r4_prof_bench -name "*kernel_stats.csv" | head -1)
