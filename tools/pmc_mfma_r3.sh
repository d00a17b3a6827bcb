#!/bin/bash
# rocprofv3 MFMA-busy counters of the fp32 MFMA kernels (similarity GEMM, grouped relation-side GEMM), one pass each.
#   usage (repo root, GPU box): bash tools/pmc_mfma_r3.sh <out_dir>
OUT=${1:-gpurun_out/pmc_mfma_r3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/sim -o p -- python3 $R/tools/simgemm_probe.py > $R/$OUT/sim.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/gg -o p -- python3 $R/tools/gg_probe.py "NN x4" > $R/$OUT/gg.log 2>&1
cd $R
python3 - <<PY
import csv, collections, json
res = {}
for tag in ("sim", "gg"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    try:
        for r in csv.DictReader(open("$OUT/%s/p_counter_collection.csv" % tag)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    except OSError as ex:
        res[tag] = {"error": str(ex)}
        continue
    for k, c in acc.items():
        if "sim_gemm" in k or "grouped_gemm" in k:
            busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / max(len(c["SQ_VALU_MFMA_BUSY_CYCLES"]), 1)
            act = sum(c["GRBM_GUI_ACTIVE"]) / max(len(c["GRBM_GUI_ACTIVE"]), 1)
            res["sim_gemm_kernel" if "sim_gemm" in k else "grouped_gemm_kernel"] = {"launches": len(c["GRBM_GUI_ACTIVE"]), "SQ_VALU_MFMA_BUSY_CYCLES_mean": busy,
                                          "GRBM_GUI_ACTIVE_mean": act, "MfmaUtil_percent_mean": busy / (act * 1024) * 100 * 8}
print(json.dumps(res, indent=1))
PY
