"""Replay-to-replay period of a hipGraph-replayed step from a rocprofv3 kernel trace: the interval between the ends of consecutive
fused-Adam kernels (one per step), and the gap between the end of a step's last kernel and the start of the next step's first.
usage: python tools/closed/step_periods.py <p_kernel_trace.csv>"""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if ('multi_tensor_apply' in r['Kernel_Name'] and 'FusedOptimizer' in r['Kernel_Name']) or 'adam_step_kernel' in r['Kernel_Name']]   # the step's last kernel: either optimizer
ends = np.array([int(rows[i]['End_Timestamp']) for i in adam], dtype=np.float64)
per = np.diff(ends) / 1e3
gaps = np.array([(int(rows[i + 1]['Start_Timestamp']) - int(rows[i]['End_Timestamp'])) / 1e3 for i in adam[:-1] if i + 1 < len(rows)])
print("steps", len(adam), "period us: median %.1f mean %.1f min %.1f max %.1f" % (np.median(per), per.mean(), per.min(), per.max()))
print("last 12 periods:", np.round(per[-12:], 1))
print("gap after Adam us: median %.1f mean %.1f max %.1f" % (np.median(gaps), gaps.mean(), gaps.max()))
big = np.flatnonzero(per > 1.05 * np.median(per))
print("periods > 1.05 x median:", len(big), "of", len(per), np.round(per[big][:10], 1))
