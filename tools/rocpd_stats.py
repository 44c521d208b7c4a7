"""Kernel statistics (and an optional per-step timeline) from a rocprofv3 results .db (rocpd sqlite schema) — the
same table `--stats` writes as CSV, for runs whose output format was left at the default.
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--csv out.csv] [--timeline]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(n, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append('"%s",%d,%d,%.1f,%.2f,%d,%d' % (n, a[0], a[1], a[1] / a[0], 100.0 * a[1] / tot, a[2], a[3]))
    if "--csv" in sys.argv:
        open(sys.argv[sys.argv.index("--csv") + 1], "w").write("\n".join(lines) + "\n")
    for l in lines[:40]:
        print(l[:230])
    if "--timeline" in sys.argv:
        rows.sort(key=lambda r: r[1])
        # last full step: from the last adam_kernel but one to the last
        adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
        if len(adam) >= 2:
            lo, hi = adam[-2] + 1, adam[-1]
            t0 = rows[lo][1]
            print("kernels in last step, span ms %.3f" % ((rows[hi][2] - t0) / 1e6))
            for n, s, e in rows[lo:hi + 1]:
                if e - s > 20000:
                    print("t=%9.1f dur=%8.1f %s" % ((s - t0) / 1e3, (e - s) / 1e3, n[:70]))


if __name__ == "__main__":
    main()
