"""Summarise a rocprofv3 --kernel-trace --stats kernel_stats.csv: top kernels, per-step time."""
import csv, glob, sys
d, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + "/**/*_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# one fused-Adam launch per optimizer step: the step count of the trace (bench.py with its kernel timing on runs an extra
# one-stream pass, so the command line's --steps/--warmup undercount)
adam = [int(r["Calls"]) for r in rows if "adam_kernel" in r["Name"]]
if adam and adam[0] != steps:
    print(f"# {adam[0]} optimizer steps in the trace (argument said {steps:g}); normalising by {adam[0]}")
    steps = float(adam[0])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel ms {tot/1e6:.2f}  per step {tot/1e6/steps:.2f} ms ({steps:g} steps incl. warmup)")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[: int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    n = r["Name"].replace("(anonymous namespace)::", "")[:80]
    print(f"{float(r['TotalDurationNs'])/1e6/steps:8.3f} ms/step {float(r['Percentage']):6.2f}% calls/step {float(r['Calls'])/steps:6.1f} avg {float(r['AverageNs'])/1e3:8.1f} us  {n}")
