"""gpurun_out/pmc_r4/summary.json (tools/pmc_collect_r4.sh) -> profiles/r4_pmc_{config4,union,ja}.json: the summary's per-kernel
counters + the workloads' algorithmic byte counts and the traffic / algorithmic ratios bench.py reads.
usage: python tools/pmc_profiles_r4.py [gpurun_out/pmc_r4] [round tag, default r4]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc_r4")
RND = sys.argv[2] if len(sys.argv) > 2 else "r4"
s = json.load(open(os.path.join(src, "summary.json")))
METHOD = ("rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE and --pmc TCC_HIT_sum TCC_MISS_sum in separate passes (tools/pmc_collect_r4.sh, "
          "summarised by tools/pmc_summarize_r4.py); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies the 128-B requests of "
          "wide coalesced reads at 64 B), calibrated on the kernels' own access pattern by the samerel runs; values in KB per launch")


def fwd_bytes(N, E, d, es, dh=None):       # E(2d s + 8) + N(2d s + 12), s = table element size, dh = stored half
    dh = dh or d
    return E * (2 * dh * es + 8) + N * (dh * es + dh * es + d * 4 + 12) if es == 2 else E * (2 * d * 4 + 8) + N * (2 * d * 4 + 12)


def first(ks, prefix):
    for k, v in ks.items():
        if k.startswith(prefix):
            return v
    return None


def dump(name, obj):
    json.dump(obj, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
    print("wrote profiles/" + name)


# ---- config 4 (byte counts as committed in the first r4 collection: bench.py's synth object uses the same figures)
N, E, d = 1_000_000, 20_000_000, 300
a_f, a_b, a_bwd = 50_572_000_000, 25_972_000_000, 99_776_000_000
c4 = {"workload": "config 4: synthetic power-law graph N=1 000 000, E=20 000 000, nr=1000, d=300 (tools/agg_sweep.py 1.0 auto 300 1; "
                  "BF16=1 for the bf16 tables, halves padded to 304)",
      "method": METHOD, "algorithmic_bytes_fwd": a_f, "algorithmic_bytes_fwd_bf16": a_b, "algorithmic_bytes_bwd": a_bwd,
      "kernels": s["c4"], "kernels_bf16": s["c4bf16"],
      "calibration": {"what": "every edge on relation 0 (DBG=samerel): the relation table is one L1-resident row, so every fetched byte "
                              "is a [Q|Z] / P / Z[i] / index byte",
                      "f32": s["c4samerel"], "bf16": s["c4bf16samerel"],
                      "expected_bytes_f32": 51_772_000_000, "expected_bytes_bf16": 26_908_000_000}}
kf, kb = first(s["c4"], "rel_attn_fwd_hw"), first(s["c4bf16"], "rel_attn_fwd_hw")
cf, cb = first(s["c4samerel"], "rel_attn_fwd_hw"), first(s["c4bf16samerel"], "rel_attn_fwd_hw")
c4["ratios"] = {"fwd_f32_traffic_over_algorithmic": kf["traffic_bytes_corrected"] / a_f,
                "fwd_bf16_traffic_over_algorithmic": kb["traffic_bytes_corrected"] / a_b,
                "calibration_f32": cf["traffic_bytes_corrected"] / 51_772_000_000 if cf else None,
                "calibration_bf16": cb["traffic_bytes_corrected"] / 26_908_000_000 if cb else None}
dump(RND + "_pmc_config4.json", c4)
if s.get("union"):
    dump(RND + "_pmc_union.json", {
        "workload": "config 3: the REAL union of the five DBP-5L KGs, N=56 589, E=197 604 (train-mode graphs), nr=4805, d=300 "
                    "(tools/union_agg_probe.py); fp32 and bf16 tables (bf16 halves padded to 304)",
        "method": METHOD, "algorithmic_bytes_fwd": 612323100, "algorithmic_bytes_fwd_bf16": 341244900,
        "note": "341-612 MB working sets: Infinity-Cache resident, traffic below the algorithmic bytes", "kernels": s["union"]})
if s.get("ja"):
    dump(RND + "_pmc_ja.json", {"workload": "the REAL DBP-5L ja KG, train-mode graph N=11 805, E=17 979, d=300 (tools/ja_sweep.py ja-real)",
                            "method": METHOD, "algorithmic_bytes_fwd": 71767092, "algorithmic_bytes_bwd": 129129912, "kernels": s["ja"]})
