# SQ counters of the conv kernels on two layer shapes (tools/conv_bench.py), three passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/tools/conv_bench.py --only ${ONLY:-fwd} --layers l1.3x3,l4.3x3 --reps 5"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pcA -o a --output-format csv -- $B > $R/gpurun_out/pcA.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $R/gpurun_out/pcB -o b --output-format csv -- $B > $R/gpurun_out/pcB.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/pcC -o c --output-format csv -- $B > $R/gpurun_out/pcC.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC -d $R/gpurun_out/pcD -o d --output-format csv -- $B > $R/gpurun_out/pcD.log 2>&1 &&
echo done
