"""GPU idle analysis from a rocprofv3 kernel trace: union of kernel intervals vs wall time over the last N steps
(steps are delimited by adam_kernel launches)."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
# step boundaries: last adam launch of each step = an adam followed by a non-adam kernel
ends = [i for k, i in enumerate(adam) if k + 1 == len(adam) or adam[k + 1] != i + 1]
if len(ends) < 3:
    sys.exit("need >= 3 steps")
for a, b in zip(ends[-3:-1], ends[-2:]):
    seg = rows[a + 1:b + 1]
    t0, t1 = seg[0][0], max(e for _, e, _ in seg)
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in seg:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    ksum = sum(e - s for s, e, _ in seg)
    gaps = sorted(((seg[i + 1][0] - max(x[1] for x in seg[:i + 1][-8:]), seg[i][2][:40], seg[i + 1][2][:40]) for i in range(len(seg) - 1)), reverse=True)[:6]
    print(f"step: wall {(t1 - t0) / 1e6:.3f} ms, busy(union) {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms, kernel-sum {ksum / 1e6:.3f} ms, launches {len(seg)}")
    for g, a_, b_ in gaps:
        print(f"   gap {g / 1e3:7.1f} us  after {a_} -> before {b_}")
