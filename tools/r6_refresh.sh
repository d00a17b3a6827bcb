#!/bin/bash
# round 6, after the last kernel change: the MFMA-busy PMC pass, the default bench line (+ bench_full.json) and the kernel-trace
# stats of the same command at HEAD (the full pass, tools/r6_final.sh, ran before the similarity GEMM's last change).
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"; mkdir -p gpurun_out
bash tools/pmc_mfma_r3.sh gpurun_out/pmc_mfma_r6 > gpurun_out/r6_pmc_mfma_raw.json 2> gpurun_out/pmc_mfma_r6.err
python3 - <<'PY'
import json
raw = json.load(open("gpurun_out/r6_pmc_mfma_raw.json"))
json.dump({"method": "tools/pmc_mfma_r3.sh (run by tools/r6_refresh.sh): rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv (one pass, no trace options); MfmaUtil = busy / (GRBM_GUI_ACTIVE/8 XCDs x 1024 SIMDs) x 100",
           "collected": "round 6 at HEAD, MI355X; sim_gemm_kernel<false, 4> (the 128 x 256 tile launch_sim picks for the 12 000^2 product), interleaved slab body, next tile's first slab staged behind the epilogue",
           "kernels": raw,
           "note": "sim_gemm: matrix pipe busy ~80 % of the launch (round 5: 74.6 %); the per-build breakdown is profiles/r6_simgemm_ablation.txt."},
          open("profiles/r6_pmc_mfma.json", "w"), indent=1)
PY
cp profiles/r6_pmc_mfma.json gpurun_out/
python bench.py > gpurun_out/r6_bench_line.json 2> gpurun_out/r6_bench.err
cp gpurun_out/bench_full.json gpurun_out/r6_bench.json
tail -c 700 gpurun_out/r6_bench_line.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6_prof_bench -o p -- python3 $R/bench.py > $R/gpurun_out/r6_bench_under_rocprof.json 2> $R/gpurun_out/r6_prof_bench.err
cd $R
find gpurun_out/r6_prof_bench -name "*kernel_trace.csv" -size +4M -delete
cp $(find gpurun_out/r6_prof_bench -name "*kernel_stats.csv" | head -1) gpurun_out/r6_bench_kernel_stats.csv
grep -E "rel_attn_fwd_kernel<3, 2, 75, float>|sim_gemm" gpurun_out/r6_bench_kernel_stats.csv | cut -c1-200
