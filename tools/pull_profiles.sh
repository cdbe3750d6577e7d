#!/bin/bash
# Copy the artefacts tools/final_profiles.sh left in gpurun_out/final/ (pulled back by gpurun) into profiles/, named per
# round:  tools/pull_profiles.sh r03 v1
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
RN=${1:?round tag, e.g. r03}; V=${2:?version tag, e.g. v1}
O=$R/gpurun_out/final; P=$R/profiles
cp $O/bench.json $P/${RN}_bench_b256_bf16_${V}.json
cp $O/kernel_stats.txt $P/${RN}_bench_b256_bf16_kernel_stats_${V}.txt
cp $O/kernel_stats_serialized.txt $P/${RN}_bench_b256_bf16_kernel_stats_${V}_serialized.txt
for d in kt kts; do
  f=$(find $O/$d -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $P/${RN}_bench_b256_bf16_kernel_stats_${V}$([ $d = kts ] && echo _serialized).csv
done
cp $O/igemm_traffic.json $P/${RN}_igemm_traffic.json
cp $O/pmc_by_kernel.txt $P/${RN}_pmc_fetch_write_by_kernel.txt
for c in cfg2:cfg2_image_only_b128_bf16 cfg5:cfg5_signal12_b512_bf16 f2:f2_fullres_250x2500_b32_bf16 cfg3_frozen:cfg3_frozen_encoders_b256_bf16 cfg3_fp32:cfg3_b256_fp32 cfg5_fp32:cfg5_signal12_b512_fp32; do
  [ -s $O/bench_${c%%:*}.json ] && cp $O/bench_${c%%:*}.json $P/${RN}_bench_${c##*:}.json
done
ls -la $P | grep ${RN}_ | wc -l
