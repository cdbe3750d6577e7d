cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --workload image_only --batch 256 --steps 2 --warmup 1 --no-cpu-baseline --no-prof"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pmcA -o a --output-format csv -- $B > $R/gpurun_out/pmcA.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS -d $R/gpurun_out/pmcB -o b --output-format csv -- $B > $R/gpurun_out/pmcB.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD -d $R/gpurun_out/pmcC -o c --output-format csv -- $B > $R/gpurun_out/pmcC.log 2>&1 &&
echo done
