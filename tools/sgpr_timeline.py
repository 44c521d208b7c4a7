"""one SGPRSS bound + gradient evaluation from a rocprofv3 --kernel-trace csv of tools/time_sgpr.py: every launch in
order (start offset, duration, gap to the previous end), delimited by sgpr_finish_kernel (one per evaluation)
    python tools/sgpr_timeline.py <kernel_trace.csv> [which-evaluation-from-the-end, default 2]"""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1]))[-1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'sgpr_finish_kernel' in r['Kernel_Name']]
ev = rows[idx[-back - 1] + 1: idx[-back] + 1]
t0 = int(ev[0]['Start_Timestamp'])
print("launches per evaluation", len(ev), "span ms", (max(int(r['End_Timestamp']) for r in ev) - t0) / 1e6)
prev_end, busy, tot = t0, 0, {}
for r in ev:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = re.sub(r"\(.*", "", r['Kernel_Name']).replace("void ", "")[:60]
    print("t=%8.1f dur=%7.1f gap=%6.1f %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name))
    prev_end = max(prev_end, e)
    tot[name] = tot.get(name, 0) + (e - s) / 1e3
print("-- by kernel (us)")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("%8.1f %s" % (v, k))
