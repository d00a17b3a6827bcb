"""rocprofv3 --pmc CSVs (scratch/pmc_collect.sh) -> per-kernel mean FETCH_SIZE / WRITE_SIZE and corrected traffic.
Corrections per MI355X_MICROARCH.md (HBM section): units are KB; on gfx950 FETCH_SIZE tallies the 128-B requests of
16-B/lane reads at 64 B -> doubled; WRITE_SIZE is exact for 16-B/lane stores and float atomics.
(bf16 tables are read 8 B per lane: uncalibrated width -- the doubled figure is an upper bound; see the note field.)"""
import csv, collections, json, sys, os
d = sys.argv[1]
def load(path):
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return out
def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0]
res = {}
for key in ("ja", "c4", "c4bf16"):
    f = load(os.path.join(d, key + "_FETCH_SIZE", "p_counter_collection.csv"))
    w = load(os.path.join(d, key + "_WRITE_SIZE", "p_counter_collection.csv"))
    ks = {}
    for name in f:
        if "rel_attn" not in name:
            continue
        fm = sum(f[name]) / len(f[name]); wm = sum(w.get(name, [0])) / max(1, len(w.get(name, [0])))
        ks[short(name)] = {"launches": len(f[name]), "FETCH_SIZE_KB_mean": fm, "WRITE_SIZE_KB_mean": wm,
                           "traffic_bytes_corrected": (2 * fm + wm) * 1024}
    res[key] = ks
print(json.dumps(res, indent=1))
