import csv, glob, sys, collections
for d in sys.argv[1:]:
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        if "igemm" not in n and "wgrad_kernel" not in n and "stem" not in n:
            continue
        key = (n[:52], r["Grid_Size"])
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(k)
        for c, v in cs.items():
            print(f"    {c:28s} {sum(v)/len(v):16.0f}")
