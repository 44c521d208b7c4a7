"""Summarise rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per dispatch."""
import csv, sys, collections, re
def short(n):
    n = re.sub(r"\(.*", "", n); n = n.replace("void ", "")
    return n[:60]
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", path)
    for k, cs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", kv[1].get("FETCH_SIZE", kv[1].get("WRITE_SIZE", [0]))))):
        print("%-62s n=%4d  " % (k, len(next(iter(cs.values())))) + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
