#!/bin/bash
# rocprofv3 SQ counters of l1_score_kernel (VALU roofline of K6).  usage (repo root, GPU box): bash tools/closed/pmc_l1_r2.sh <out_dir>
OUT=${1:-gpurun_out/pmc_l1}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/sq -o p -- python3 $R/tools/closed/l1_probe.py > $R/$OUT/sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace -o p -- python3 $R/tools/closed/l1_probe.py > $R/$OUT/trace.log 2>&1
ls $R/$OUT/sq $R/$OUT/trace | head
