#!/bin/bash
# round-6 evidence at HEAD, one GPU-box pass: PMC passes (aggregation kernels on config 4 / the real union / the real ja graph; MFMA
# busy of the similarity GEMM), the whole GPU suite, the default bench line (+ bench_full.json), kernel-trace stats of the same
# command, the three step breakdowns.  Outputs under gpurun_out/; the summaries are copied into profiles/ afterwards.
#   usage (repo root, GPU box): bash tools/r6_final.sh [pmc]
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; mkdir -p gpurun_out
if [ "$1" = "pmc" ]; then
  bash tools/pmc_collect_r4.sh gpurun_out/pmc_r6 > gpurun_out/pmc_r6.log 2>&1
  python3 tools/pmc_profiles_r4.py gpurun_out/pmc_r6 r6 >> gpurun_out/pmc_r6.log 2>&1
  cp profiles/r6_pmc_config4.json profiles/r6_pmc_union.json profiles/r6_pmc_ja.json gpurun_out/ 2>/dev/null
  bash tools/pmc_mfma_r3.sh gpurun_out/pmc_mfma_r6 > gpurun_out/r6_pmc_mfma_raw.json 2> gpurun_out/pmc_mfma_r6.err
fi
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r6_gpu_tests.log
cat gpurun_out/r6_gpu_tests.log
python bench.py > gpurun_out/r6_bench_line.json 2> gpurun_out/r6_bench.err
cp gpurun_out/bench_full.json gpurun_out/r6_bench.json
tail -c 400 gpurun_out/r6_bench_line.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6_prof_bench -o p -- python3 $R/bench.py > $R/gpurun_out/r6_bench_under_rocprof.json 2> $R/gpurun_out/r6_prof_bench.err
cd $R
find gpurun_out/r6_prof_bench -name "*kernel_trace.csv" -size +4M -delete
cp $(find gpurun_out/r6_prof_bench -name "*kernel_stats.csv" | head -1) gpurun_out/r6_bench_kernel_stats.csv
head -8 gpurun_out/r6_bench_kernel_stats.csv
bash tools/r6_steps.sh
