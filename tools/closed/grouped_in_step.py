"""Where the grouped relation-side launches sit inside the last profiled step (rocprofv3 kernel trace CSV), and which
kernels they overlap with in time (the side stream / hipGraph branch really running next to the node side)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'multi_tensor_apply' in r['Kernel_Name'] and 'FusedOptimizer' in r['Kernel_Name']]
a, b = adam[-2], adam[-1]
win = rows[a + 1:b + 1]
t0 = int(win[0]['Start_Timestamp'])
print("step span %.1f us" % ((int(win[-1]['End_Timestamp']) - t0) / 1e3))
for r in win:
    n = r['Kernel_Name']
    if 'grouped_gemm' in n:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        ov = [(min(e, int(o['End_Timestamp'])) - max(s, int(o['Start_Timestamp'])), o['Kernel_Name']) for o in win
              if o is not r and int(o['Start_Timestamp']) < e and int(o['End_Timestamp']) > s]
        ovs = ", ".join("%s %.1f us" % (k.split("(")[0][-28:], d / 1e3) for d, k in ov)
        print("grouped at %8.1f us  dur %6.1f us  grid %sx%s  overlaps: %s" % (
            (s - t0) / 1e3, (e - s) / 1e3, r.get('Grid_Size_X', '?'), r.get('Grid_Size_Y', '?'), ovs or "-"))
