#!/bin/bash
# round-5 evidence at HEAD, one GPU-box pass: PMC passes (aggregation kernels on config 4 / the real union / the real ja graph; MFMA
# busy of the similarity GEMM), the whole GPU suite, the default bench line, kernel-trace stats of the same command, step breakdowns
# of the headline ja step and the el + ja pair step.  Outputs under gpurun_out/; copy the summaries into profiles/ afterwards
# (tools/r5_collect.sh does).     usage (repo root, GPU box): bash tools/r5_final.sh [pmc]
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; mkdir -p gpurun_out
if [ "$1" = "pmc" ]; then
  bash tools/pmc_collect_r4.sh gpurun_out/pmc_r5 > gpurun_out/pmc_r5.log 2>&1
  python3 tools/pmc_profiles_r4.py gpurun_out/pmc_r5 r5 >> gpurun_out/pmc_r5.log 2>&1
  cp profiles/r5_pmc_config4.json profiles/r5_pmc_union.json profiles/r5_pmc_ja.json gpurun_out/ 2>/dev/null
  bash tools/pmc_mfma_r3.sh gpurun_out/pmc_mfma_r5 > gpurun_out/r5_pmc_mfma_raw.json 2> gpurun_out/pmc_mfma_r5.err
fi
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r5_gpu_tests.log
cat gpurun_out/r5_gpu_tests.log
python bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err
tail -c 600 gpurun_out/r5_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5_prof_bench -o p -- python3 $R/bench.py > $R/gpurun_out/r5_bench_under_rocprof.json 2> $R/gpurun_out/r5_prof_bench.err
cd $R
find gpurun_out/r5_prof_bench -name "*kernel_trace.csv" -size +4M -delete
cp $(find gpurun_out/r5_prof_bench -name "*kernel_stats.csv" | head -1) gpurun_out/r5_bench_kernel_stats.csv
head -8 gpurun_out/r5_bench_kernel_stats.csv
bash tools/step_profile2.sh gpurun_out/r5_step_ja --ja > gpurun_out/r5_step_ja.log 2>&1
bash tools/step_profile2.sh gpurun_out/r5_step_pair --batched 1 > gpurun_out/r5_step_pair.log 2>&1
{ echo "== real DBP-5L ja single-KG step (hipGraph replay; tools/step_profile2.sh --ja)"; cat gpurun_out/r5_step_ja/step_breakdown.txt; cat gpurun_out/r5_step_ja/probe.json;
  echo; echo "== real el + ja pair step, batched (tools/step_profile2.sh --batched 1)"; cat gpurun_out/r5_step_pair/step_breakdown.txt; cat gpurun_out/r5_step_pair/probe.json; } > gpurun_out/r5_step_breakdown.txt
head -4 gpurun_out/r5_step_breakdown.txt
