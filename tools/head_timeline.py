"""head of one bench step (until A = W Kuf starts) from a rocprofv3 --kernel-trace csv"""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1]))[-1]
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
step = rows[idx[-3] + 1: idx[-2] + 1]; t0 = int(rows[idx[-3]]['End_Timestamp'])
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = re.sub(r"\(.*", "", r['Kernel_Name']).replace("void ", "")[:46]
    if (e - s) / 1e3 > 25:
        print("t=%7.1f dur=%7.1f end=%7.1f %-46s q=%s" % ((s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3, nm, r['Queue_Id']))
    if 'false, false, 1>' in nm: break
print("step span:", (int(step[-1]['End_Timestamp']) - t0) / 1e3)
