"""Per-kernel-name averages of rocprofv3 --pmc counter_collection.csv (optionally filtered by substring)."""
import csv, glob, sys, collections
d = sys.argv[1]; filt = sys.argv[2:] or [""]
f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    if not any(x in n for x in filt):
        continue
    agg[n[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    print(k, " ".join(f"{c}: n={len(v)} avg={sum(v)/len(v):.4g} sum={sum(v):.4g}" for c, v in cs.items()))
