"""Timeline of ONE replayed step from a rocprofv3 kernel trace (csv): start offset, duration, queue, kernel -- to see what overlaps.
usage: python tools/closed/step_timeline.py <p_kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if ('multi_tensor_apply' in r['Kernel_Name'] and 'FusedOptimizer' in r['Kernel_Name']) or 'adam_step_kernel' in r['Kernel_Name']]   # the step's last kernel: either optimizer
a, b = adam[-2], adam[-1]
t0 = int(rows[a]['End_Timestamp'])
busy = 0
last_end = t0
idle = 0
for r in rows[a + 1:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    busy += e - s
    if s > last_end:
        idle += s - last_end
    last_end = max(last_end, e)
    print("%8.1f %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'].replace('(anonymous namespace)::', '')[:90]))
print("span us %.1f  sum of kernels us %.1f  idle us %.1f" % ((last_end - t0) / 1e3, busy / 1e3, idle / 1e3))
