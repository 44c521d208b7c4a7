"""one bench step from a rocprofv3 --kernel-trace csv: span, idle time, the long kernels in order"""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1]))[-1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
step = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(step[0]['Start_Timestamp'])
print("kernels in step", len(step), "span ms", (max(int(r['End_Timestamp']) for r in step) - t0) / 1e6)
busy_end, idle = t0, 0
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s > busy_end:
        idle += s - busy_end
    busy_end = max(busy_end, e)
    if (e - s) / 1e3 >= thr:
        name = re.sub(r"\(.*", "", r['Kernel_Name']).replace("void ", "")[:46]
        print("t=%8.1f dur=%8.1f %-48s q=%s" % ((s - t0) / 1e3, (e - s) / 1e3, name, r['Queue_Id']))
print("idle (no kernel running) us:", idle / 1e3)
