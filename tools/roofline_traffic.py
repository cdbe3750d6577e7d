"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of
`bench.py --no-prof` into the per-launch HBM traffic of the implicit-GEMM conv kernel class (igemm_kernel and conv_halo_kernel, bf16 instantiations).
Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half the
bytes of wide (16 B/lane) coalesced reads -- the kernel's reads are 16-B LDS-DMA loads -> x2;
WRITE_SIZE is exact for its stores.  Units of both counters: KiB.
usage: roofline_traffic.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, sys

def collect(d, counter):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and ("igemm_kernel<unsigned short" in r["Kernel_Name"] or "conv_halo_kernel<" in r["Kernel_Name"]):
            tot += float(r["Counter_Value"]); n += 1
    return tot, n

fetch, nf = collect(sys.argv[1], "FETCH_SIZE")
write, nw = collect(sys.argv[2], "WRITE_SIZE")
assert nf == nw and nf > 0, (nf, nw)
per_launch = (2.0 * fetch + write) * 1024.0 / nf
out = {"kernel": "igemm_kernel<bf16> + conv_halo_kernel (conv fwd + dgrad)", "launches_profiled": nf,
       "fetch_size_kib_sum": fetch, "write_size_kib_sum": write, "fetch_correction": 2.0,
       "traffic_bytes_per_launch": per_launch,
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc_traffic.sh) on "
                 "`python bench.py --steps 2 --warmup 1 --no-prof --no-cpu-baseline --serialize`"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
