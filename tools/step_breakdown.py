"""Per-step kernel-time breakdown of a hipGraph-replayed step from a rocprofv3 --kernel-trace CSV (the last 20 steps, delimited by
the optimizer's kernel).   usage: step_breakdown.py <kernel_trace.csv> [top-n] [grid] [--json <out.json> <workload-key>]
--json merges {"workloads": {<key>: {span_us, busy_us, kernels_per_step, groups, kernels: [{name, us_per_step, calls_per_step}]}}}
into <out.json> (profiles/r6_step_kernels.json: what bench.py's roofline.frac_in_step reads)."""
import csv, collections, json, os, sys
JSON_OUT = None
if "--json" in sys.argv:
    i = sys.argv.index("--json")
    JSON_OUT = (sys.argv[i + 1], sys.argv[i + 2])
    del sys.argv[i:i + 3]
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
adam=[i for i,r in enumerate(rows) if ('multi_tensor_apply' in r['Kernel_Name'] and 'FusedOptimizer' in r['Kernel_Name']) or 'adam_step_kernel' in r['Kernel_Name']]   # the step's last kernel: either optimizer
a,b=adam[-21],adam[-1]
win=rows[a+1:b+1]
tot=collections.Counter(); cnt=collections.Counter()
for r in win:
    n=r['Kernel_Name']; d=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    tot[n]+=d; cnt[n]+=1
busy=sum(tot.values())/20
span=(int(rows[b]['End_Timestamp'])-int(rows[a]['End_Timestamp']))/20
print("span us %.1f busy us %.1f kernels/step %.1f"%(span/1e3, busy/1e3,len(win)/20))
groups=collections.Counter()
for n,t in tot.items():
    g = "gemm(lib)" if n.startswith("Cijk") else ("jmac" if "anonymous namespace" in n and "at::native" not in n else "torch-elementwise")
    groups[g]+=t/20/1e3
print(dict(groups))
for n,t in tot.most_common(int(sys.argv[2]) if len(sys.argv)>2 else 30):
    print("%8.1f us  x%5.1f  %s"%(t/20/1e3,cnt[n]/20,n[:110]))
if len(sys.argv) > 3 and sys.argv[3] == "grid":       # launch geometry per kernel: workgroups, workgroup size, VGPRs, LDS
    seen = {}
    for r in win:
        n = r['Kernel_Name']
        g = (int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])) // max(1, int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z']))
        seen.setdefault((n, g, int(r['Workgroup_Size_X']), int(r['VGPR_Count']), int(r['LDS_Block_Size'])), []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    print("-- launch geometry: us (mean)  workgroups  wg-size  vgpr  lds  kernel")
    for (n, g, wg, vg, lds), d in sorted(seen.items(), key=lambda kv: -sum(kv[1])):
        if not n.startswith("Cijk"):
            print("%8.1f  x%4.1f  %7d  %5d  %4d  %6d  %s" % (sum(d) / len(d) / 1e3, len(d) / 20, g, wg, vg, lds, n[:90]))
if JSON_OUT:
    fn, key = JSON_OUT
    doc = json.load(open(fn)) if os.path.exists(fn) else {}
    doc.setdefault("source", "rocprofv3 --kernel-trace of tools/pair_probe.py (hipGraph replay), last 20 steps; tools/step_breakdown.py --json")
    doc.setdefault("workloads", {})[key] = {
        "span_us": span / 1e3, "busy_us": busy / 1e3, "kernels_per_step": len(win) / 20, "groups_us": dict(groups),
        "kernels": [{"name": n, "us_per_step": t / 20 / 1e3, "calls_per_step": cnt[n] / 20} for n, t in tot.most_common()]}
    json.dump(doc, open(fn, "w"), indent=1)
