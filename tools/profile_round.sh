#!/bin/bash
# Collects the rocprofv3 evidence of a round on the MI355X box into gpurun_out/$1/.
#   bash tools/profile_round.sh r03
set -o pipefail
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== bench" && timeout -k 10 400 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 600 $out/bench_default.json
echo "== kernel stats of the bench step"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_stats -o b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_stats.log 2>&1
cp $out/bench_stats/b_kernel_stats.csv $out/bench_kernel_stats.csv 2>/dev/null
run_pmc() {  # name, counters..., then -- command
  name=$1; shift
  ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d $out/pmc_$name -o p -- "$@" > $out/pmc_$name.log 2>&1
}
echo "== PMC passes, fused FC kernel f16x3"
run_pmc fc1 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -- python3 tools/run_inference.py --precision f16x3 --reps 2
run_pmc fc2 FETCH_SIZE -- python3 tools/run_inference.py --precision f16x3 --reps 2
run_pmc fc3 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- python3 tools/run_inference.py --precision f16x3 --reps 2
run_pmc fc4 SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -- python3 tools/run_inference.py --precision f16x3 --reps 2
python3 tools/pmc_summary.py $out fused_fista_kernel > $out/fused_f16x3_pmc.txt; cat $out/fused_f16x3_pmc.txt
echo "== secondary configs"
timeout -k 10 300 python3 tools/run_configs.py > $out/secondary_configs.txt 2>&1; grep -v amdgpu $out/secondary_configs.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cfg_stats -o c -- python3 tools/run_configs.py > /dev/null 2>&1
cp $out/cfg_stats/c_kernel_stats.csv $out/secondary_kernel_stats.csv 2>/dev/null
timeout -k 10 300 python3 tools/run_configs.py conv_geometries 2>&1 | grep -v amdgpu > $out/conv_geometries.txt; cat $out/conv_geometries.txt
echo "== PMC passes, streamed kernel (subspace configs[3], 50 iterations)"
run_pmc st1 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -- python3 tools/run_subspace_once.py
run_pmc st2 FETCH_SIZE -- python3 tools/run_subspace_once.py
run_pmc st3 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- python3 tools/run_subspace_once.py
python3 tools/pmc_summary.py $out fused_stream_kernel > $out/stream_pmc.txt; cat $out/stream_pmc.txt
echo "== PMC passes, fused conv kernel (configs[4], b = 8, 20 iterations)"
run_pmc cv1 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -- python3 tools/run_configs.py conv
run_pmc cv2 FETCH_SIZE -- python3 tools/run_configs.py conv
run_pmc cv3 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- python3 tools/run_configs.py conv
run_pmc cv4 SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS -- python3 tools/run_configs.py conv
python3 tools/pmc_summary.py $out conv_ > $out/conv_pmc.txt; cat $out/conv_pmc.txt
VTC_CONV_STAMPS=1 timeout -k 10 100 python3 tools/run_configs.py conv 2>&1 | grep -A10 "do_synth=1" | tail -11 > $out/conv_stamps.txt; cat $out/conv_stamps.txt
[ -x tools/micro/lds_unaligned ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/lds_unaligned.hip -o tools/micro/lds_unaligned 2>/dev/null
timeout -k 10 60 tools/micro/lds_unaligned > $out/lds_unaligned.txt 2>&1; cat $out/lds_unaligned.txt
[ -x tools/peaks/peaks ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/peaks/peaks.hip -o tools/peaks/peaks 2>/dev/null
timeout -k 10 120 tools/peaks/peaks > $out/peaks.txt 2>&1; grep -i "HBM\|MFMA\|stream" $out/peaks.txt
[ -x tools/micro/stream_mfma ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/stream_mfma.hip -o tools/micro/stream_mfma 2>/dev/null
timeout -k 10 60 tools/micro/stream_mfma > $out/stream_mfma.txt 2>&1; cat $out/stream_mfma.txt
timeout -k 10 300 python3 tools/precision_conv_report.py 2>&1 | grep -v amdgpu > $out/precision_conv.txt; tail -8 $out/precision_conv.txt
timeout -k 10 300 python3 tools/precision_report.py 2>&1 | grep -v amdgpu > $out/precision_fc.txt; tail -12 $out/precision_fc.txt
echo "== fully-connected shapes outside the headline kernel, reproducibility soak"
timeout -k 10 300 python3 tools/time_fc_shapes.py 2>&1 | grep -v amdgpu > $out/fc_other_shapes.txt; cat $out/fc_other_shapes.txt
timeout -k 10 600 python3 tools/soak_reproducibility.py 150 2>&1 | grep -v amdgpu > $out/soak_reproducibility.txt; tail -2 $out/soak_reproducibility.txt
echo "== example sizes"
timeout -k 10 200 python3 tools/time_fc_example.py 2>&1 | grep -v amdgpu > $out/example_sizes.txt
timeout -k 10 200 python3 tools/time_conv_example.py 2>&1 | grep -v amdgpu >> $out/example_sizes.txt; cat $out/example_sizes.txt
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ex_stats -o e -- python3 tools/time_fc_example.py > /dev/null 2>&1
cp $out/ex_stats/e_kernel_stats.csv $out/fc_example_kernel_stats.csv 2>/dev/null
rm -rf $out/bench_stats $out/cfg_stats $out/ex_stats
echo done
