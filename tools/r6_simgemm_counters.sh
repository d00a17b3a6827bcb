#!/bin/bash
# round 6: where the similarity GEMM's non-MFMA time goes, by counter (the product build, quality shape, 12 launches): LDS bank
# conflicts against LDS-active cycles, wave-cycles waiting for any instruction / for LDS, texture-addresser busy, VMEM issue cycles.
# One counter group per pass (no trace options).   usage (GPU box, repo root): bash tools/r6_simgemm_counters.sh
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$R"; mkdir -p gpurun_out/r6_sg_ctr
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_BUSY_CU_CYCLES" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL TCP_TA_DATA_STALL_CYCLES"; do
  tag=$(echo $grp | tr ' ' '+')
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/r6_sg_ctr/$tag -o p -- python3 $R/tools/simgemm_probe.py > $R/gpurun_out/r6_sg_ctr/$tag.log 2>&1 )
done
python3 - <<'PY' > gpurun_out/r6_simgemm_counters.json
import csv, collections, glob, json
out = {}
for f in glob.glob("gpurun_out/r6_sg_ctr/*/p_counter_collection.csv"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "sim_gemm" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k] = sum(v) / len(v)
print(json.dumps(out, indent=1))
PY
cat gpurun_out/r6_simgemm_counters.json
find gpurun_out/r6_sg_ctr -name "*.csv" -size +1M -delete
