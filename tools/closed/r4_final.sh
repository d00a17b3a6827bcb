#!/bin/bash
# round-4 evidence at HEAD: PMC passes, GPU suite, default bench line, kernel-trace stats of the bench, step breakdowns (ja / pair)
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
bash tools/pmc_collect_r4.sh gpurun_out/pmc_r4 > gpurun_out/pmc_r4.log 2>&1
python3 tools/pmc_profiles_r4.py gpurun_out/pmc_r4
bash tools/closed/r4_check.sh
bash tools/step_profile2.sh gpurun_out/r4_step_ja --ja
bash tools/step_profile2.sh gpurun_out/r4_step_pair --batched 1
cp profiles/r4_pmc_*.json gpurun_out/      # (written on the GPU box: copy them back into profiles/ in the tree afterwards)
