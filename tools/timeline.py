"""Per-stream timeline of one training step from a rocprofv3 --kernel-trace CSV: for every stream (queue) the busy time,
the gaps, and the chronological kernel list of the LAST complete step (between two fused-Adam launches).
Usage: python tools/timeline.py <dir-with-*kernel_trace.csv> [--list]"""
import csv, glob, sys, collections
d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
adam = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
lo, hi = adam[-3], adam[-2]          # one whole step between two Adam launches (2 runs of Adam per step: take spaced ones)
step = rows[lo + 1:hi + 1]
t0, t1 = step[0]["s"], step[-1]["e"]
print(f"step wall {(t1 - t0) / 1e3:.1f} us, {len(step)} launches")
byq = collections.defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
for q, ks in byq.items():
    busy = sum(k["e"] - k["s"] for k in ks)
    gaps = sum(max(0, b["s"] - a["e"]) for a, b in zip(ks, ks[1:]))
    print(f"queue {q}: {len(ks)} launches, busy {busy / 1e3:.1f} us, gaps between its kernels {gaps / 1e3:.1f} us, span {(ks[-1]['e'] - ks[0]['s']) / 1e3:.1f} us")
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:70]
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    a = agg[(r["Queue_Id"], short(r["Kernel_Name"]))]
    a[0] += 1; a[1] += r["e"] - r["s"]
for (q, n), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"  q{q} {t / 1e3:8.1f} us {c:4d}x  {n}")
if "--list" in sys.argv:
    for r in step:
        print(f"{(r['s'] - t0) / 1e3:9.1f} {(r['e'] - r['s']) / 1e3:7.1f} q{r['Queue_Id']} {short(r['Kernel_Name'])}")
