#!/bin/bash
# Profiles of one round, all from the same workload (tools/pmc_predict.py = the bench's 2D 2048^2 and 3D 256^3 predict_instances):
#   0. unprofiled warm run
#   1. rocprofv3 --kernel-trace (per-kernel durations of SD_PMC_STEPS steps + calibration, no trial kernels)
#   2./3. --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes: they do not fit one)
#   4. --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 (MFMA utilisation of the convolutions)
# usage (GPU box): tools/profile_round.sh r03      -> gpurun_out/r03/*.md, pair_kernel_traffic.json
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1
O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
export SD_PMC_STEPS=3
python $R/tools/pmc_predict.py > $O/warm.log 2>&1
# 1. the bench command itself under the kernel trace (its JSON line is kept next to the table: the per-kernel averages must agree with it)
rm -rf /tmp/prof_kt; (cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python bench.py --no-cpu-baseline --no-sharded --no-split-leg --steps 5 --warmup 2 > $O/bench_under_trace.log 2>&1)
db=$(find /tmp/prof_kt -name '*.db' | head -1)
python $R/tools/rocpd_summary.py $db --md > $O/kernel_stats.md 2>&1
grep "^{" $O/bench_under_trace.log > $O/bench_under_trace.json
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pmc_$i -o p -- python $R/tools/pmc_predict.py > $O/pmc_$i.log 2>&1
done
f1=$(find /tmp/pmc_1 -name 'p_counter_collection.csv' | head -1); f2=$(find /tmp/pmc_2 -name 'p_counter_collection.csv' | head -1); f3=$(find /tmp/pmc_3 -name 'p_counter_collection.csv' | head -1)
python $R/tools/profile_traffic.py ${f1%_counter_collection.csv} ${f2%_counter_collection.csv} $O/pmc_1.log $O/pair_kernel_traffic.json $O/pmc_hbm_traffic.md $tag > $O/traffic.log 2>&1
python $R/tools/pmc_multi.py ${f3%_counter_collection.csv} k_ > $O/pmc_mfma.md 2>&1
f4=$(find /tmp/pmc_4 -name 'p_counter_collection.csv' | head -1)
[ -n "$f4" ] && python $R/tools/pmc_multi.py ${f4%_counter_collection.csv} k_conv3 > $O/pmc_conv_stalls.md 2>&1
# 5. the convolution kernel per forward pass: durations (kernel trace) and HBM traffic (two PMC passes) of the K timed forward passes of
#    tools/pmc_forward.py, 2D and 3D -> conv_kernel_traffic.json (bench.py: roofline_convs.traffic) + conv_forward.md
rm -f $O/conv_kernel_traffic.json $O/conv_forward.md
for W in 2d 3d; do
  for P in KT FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/cf_$P
    if [ $P = KT ]; then timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/cf_$P -o p -- python $R/tools/pmc_forward.py $W 5 > $O/conv_forward_$W.log 2>&1
    else timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/cf_$P -o p -- python $R/tools/pmc_forward.py $W 5 > $O/conv_forward_${W}_$P.log 2>&1; fi
  done
  k=$(find /tmp/cf_KT -name 'p_kernel_trace.csv' | head -1); f=$(find /tmp/cf_FETCH_SIZE -name 'p_counter_collection.csv' | head -1); w=$(find /tmp/cf_WRITE_SIZE -name 'p_counter_collection.csv' | head -1)
  python $R/tools/conv_traffic.py $W ${k%_kernel_trace.csv} ${f%_counter_collection.csv} ${w%_counter_collection.csv} $O/conv_forward_$W.log $O/conv_kernel_traffic.json $O/conv_forward.md >> $O/traffic.log 2>&1
done
head -40 $O/kernel_stats.md; cat $O/traffic.log | cut -c1-600
