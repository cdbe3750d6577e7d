# SQ counters of the round-3 conv / weight-gradient kernels on three layer shapes (tools/conv_bench.py --wgrows, batch 256), three
# separate --pmc passes (kernel-trace only).  Summary: gpurun_out/r3_pmc_conv.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/tools/conv_bench.py --wgrows --only fwd,dgrad,wgrad,dgradadd --layers l1.3x3,l2.3x3,l4.3x3 --reps 3"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $R/gpurun_out/p3A -o a --output-format csv -- $B > $R/gpurun_out/p3A.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU -d $R/gpurun_out/p3B -o b --output-format csv -- $B > $R/gpurun_out/p3B.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $R/gpurun_out/p3C -o c --output-format csv -- $B > $R/gpurun_out/p3C.log 2>&1 &&
python3 - $R/gpurun_out/p3A $R/gpurun_out/p3B $R/gpurun_out/p3C > $R/gpurun_out/r3_pmc_conv.txt <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        if not any(x in n for x in ("conv_halo_kernel", "wgrad_ring_kernel", "igemm_kernel")):
            continue
        key = (n[:64], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("LDS_Block_Size", ""))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# per launch averages; VALU/MFMA = vector instructions issued per MFMA (the guide: 2 per v_mfma_f32_16x16x32_bf16 are free);")
print("# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES (fraction of the SQ-busy cycles with a matrix pipe busy); LDS conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE")
for k, c in sorted(agg.items()):
    a = {m: sum(v) / len(v) for m, v in c.items()}
    g = lambda m: a.get(m, float('nan'))
    print(f"{k[0]:64s} grid {k[1]:>8s}  n={len(c.get('SQ_INSTS_MFMA', []))}  MFMA insts {g('SQ_INSTS_MFMA'):.3g}  VALU/MFMA {g('SQ_INSTS_VALU') / g('SQ_INSTS_MFMA'):5.2f}  LDS/MFMA {g('SQ_INSTS_LDS') / g('SQ_INSTS_MFMA'):5.2f}  SALU/MFMA {g('SQ_INSTS_SALU') / g('SQ_INSTS_MFMA'):5.2f}  "
          f"MFMA busy {g('SQ_VALU_MFMA_BUSY_CYCLES') / g('SQ_BUSY_CYCLES'):6.3f}  LDS conflict {g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE'):6.3f}")
PY
cat $R/gpurun_out/r3_pmc_conv.txt
