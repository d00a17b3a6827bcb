#!/bin/bash
# round 6: where does the similarity GEMM's matrix-pipe idle time go?  Ablation / experiment builds of sim_gemm_kernel
# (score.hip: JMAC_SG_ABLATE bits, JMAC_SG_FORCE_WJ, JMAC_SG_GLDS), timed on the config-5 shapes beside the library's
# fp32 NT GEMM (torch.mm -> hipBLASLt / rocBLAS), plus one PMC pass (MFMA busy, GRBM_GUI_ACTIVE) per build.
#   build (dev container, repo root):  bash tools/r6_simgemm_ablation.sh build     -> build/variants/jmac_<name>.so (travel with gpurun)
#   run   (GPU box, repo root):        bash tools/r6_simgemm_ablation.sh run [pmc] -> gpurun_out/r6_simgemm_ablation.txt
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
# variants: the product build ("auto": launch_sim picks the tile per shape), each tile forced, the ablations of the 128 x 128 form
# (JMAC_SG_ABLATE bits: 1 no C store, 2 no global loads after the first slab, 4 no LDS traffic / barriers in the K loop), the closed
# LDS-DMA form.  Round-6 history (profiles/r6_simgemm_ablation.txt) also holds the builds that were measured and removed: the
# round-5 loop ("burst": sixteen MFMAs, then all LDS / memory operations), s_setprio ramps, non-temporal C stores, launch bounds 2.
VARIANTS=${VARIANTS:-"auto: wj2:-DJMAC_SG_FORCE_WJ=2 wj4:-DJMAC_SG_FORCE_WJ=4 nostore:-DJMAC_SG_FORCE_WJ=2,-DJMAC_SG_ABLATE=1 noload:-DJMAC_SG_FORCE_WJ=2,-DJMAC_SG_ABLATE=2 nostore_noload:-DJMAC_SG_FORCE_WJ=2,-DJMAC_SG_ABLATE=3 mfma_only:-DJMAC_SG_FORCE_WJ=2,-DJMAC_SG_ABLATE=7 mfma_only_io:-DJMAC_SG_FORCE_WJ=2,-DJMAC_SG_ABLATE=4 glds:-DJMAC_SG_GLDS=1"}
if [ "$1" = "build" ]; then
  make -s -C jmac_amd/csrc >/dev/null || exit 1
  mkdir -p build/variants
  for v in $VARIANTS; do
    name=${v%%:*}; flags=$(echo "${v#*:}" | tr ',' ' ')
    ( mkdir -p build/variants/o_$name
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Ijmac_amd/csrc $flags -c jmac_amd/csrc/score.hip -o build/variants/o_$name/score.o &&
      /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls build/*.o | grep -v -e score.o -e aggregate_testing.o -e gemm3.o) build/variants/o_$name/score.o -o build/variants/jmac_$name.so &&
      echo "built build/variants/jmac_$name.so ($flags)" ) &
    while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
  done
  wait; exit 0
fi
mkdir -p gpurun_out
OUT=gpurun_out/r6_simgemm_ablation.txt
{ echo "# sim_gemm_kernel ablation (tools/r6_simgemm_ablation.sh run): ms per launch (HIP events, 20 launches after 3), TFLOP/s on 2 M N d"
  python3 tools/r6_simgemm_probe.py --lib-bar
  for v in $VARIANTS; do
    name=${v%%:*}
    JMAC_LIB_PATH=$R/build/variants/jmac_$name.so python3 tools/r6_simgemm_probe.py --name $name
  done; } > $OUT 2> gpurun_out/r6_simgemm_ablation.err
if [ "$2" = "pmc" ]; then
  mkdir -p gpurun_out/r6_sg_pmc
  for v in $VARIANTS; do
    name=${v%%:*}
    ( cd /tmp && export TMPDIR=/tmp && JMAC_LIB_PATH=$R/build/variants/jmac_$name.so timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r6_sg_pmc/$name -o p -- python3 $R/tools/simgemm_probe.py > $R/gpurun_out/r6_sg_pmc/$name.log 2>&1 )
    python3 - "$name" >> $OUT <<'PY'
import csv, collections, json, sys
name = sys.argv[1]
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open("gpurun_out/r6_sg_pmc/%s/p_counter_collection.csv" % name)):
        if "sim_gemm" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    act = m["GRBM_GUI_ACTIVE"] / 8
    print(json.dumps({"pmc": name, "launches": len(acc["GRBM_GUI_ACTIVE"]), "cycles_per_xcd": act, "mfma_busy_cycles_per_simd": m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024,
                      "mfma_busy_frac": m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / act, "sq_busy_cycles": m.get("SQ_BUSY_CYCLES")}))
except Exception as ex:
    print(json.dumps({"pmc": name, "error": str(ex)}))
PY
  done
  find gpurun_out/r6_sg_pmc -name "*.csv" -size +1M -delete
fi
cat $OUT
