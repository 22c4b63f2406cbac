#!/bin/bash
# Dev tool (GPU box): everything the round's DESIGN / profiles/ quote, in one call.  Outputs under gpurun_out/r03/ (copied to profiles/ by hand).
# usage: bash scripts/round_artifacts.sh [A|B]   (A: bench lines + rocprofv3 passes; B: A/B probes, microbenchmarks, static checks + placement sweep;
#        needs the experiments library and the placement-pad libraries: make experiments; placement_sweep.sh build 7; each part fits one 20-minute call)
part=${1:-AB}
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
EXP=anncur_amd/lib/libanncur_hip_exp.so
if [[ $part == *A* ]]; then
echo "== bench cfg2"; python3 bench.py > $out/bench_cfg2_n1.json 2> $out/bench_cfg2.err
echo "== bench cfg4 per gpu"; python3 bench.py --config cfg4_per_gpu --no-ivf > $out/bench_cfg4_per_gpu_n1.json 2> $out/bench_cfg4.err
echo "== bench 2 ranks gloo"; python3 bench.py --gpus 2 --backend gloo --share-gpu --steps 5 --warmup 2 --sustained-seconds 0 > $out/bench_cfg4_2ranks_gloo_one_gpu_rehearsal.json 2> $out/bench_2r.err
echo "== profile cfg2"; bash scripts/profile_round.sh r03_cfg2 cfg2 > $out/profile_cfg2.log 2>&1
echo "== profile cfg4"; bash scripts/profile_round.sh r03_cfg4 cfg4_per_gpu > $out/profile_cfg4.log 2>&1
fi
if [[ $part == *B* ]]; then
echo "== variant kernels under rocprof"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/variants_trace -o run -- python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -k "mfma16 or regression_dense or local_overflow or variant" > $out/variants_pytest.log 2>&1
f=$(find $out/variants_trace -name "*kernel_stats.csv" | head -1); grep -E "score16_kernel|score_kernel<(64|128|256), 1, 16, (false|true), false, 1>|score_kernel<(64|128|256), 1, 16, (false|true), false, 2>" $f | cut -c1-160 > $out/variant_kernels_seen.txt; cat $out/variant_kernels_seen.txt
echo "== A/B variants (experiments library, one process, interleaved)"
STAGE_PROBE_ONLY="default;mfma32;qt1;static;1 stage;f=.30;f=.15;f=.06,.30;bare;bare mfma32;bare qt1;static bare" ANNCUR_LIB=$EXP python3 scripts/stage_probe.py 100 9 2>&1 | grep -v -e warm-up -e amdgpu.ids > $out/ab_variants.txt; cat $out/ab_variants.txt
echo "== in-kernel clock"; (ANNCUR_LIB=$EXP ANNCUR_CLOCK_DETAIL=1 python3 scripts/inkernel_clock.py 256; ANNCUR_LIB=$EXP python3 scripts/inkernel_clock.py 512) 2>&1 | grep -v amdgpu.ids > $out/inkernel_clock.txt; cat $out/inkernel_clock.txt
echo "== sweep phases"; ANNCUR_LIB=$EXP python3 scripts/sweep_phases.py 2>&1 | grep -v amdgpu.ids > $out/sweep_phases.txt; cat $out/sweep_phases.txt
echo "== microbench k100/k500/k1000"
(for k in 100 500 1000; do echo "k_retvr = $k"; python3 scripts/fused_microbench.py --k $k --scan 0 --by-norm 1; done; echo "wide kernel"; python3 scripts/fused_microbench.py --K 1024 --scan 0; python3 scripts/fused_microbench.py --K 768 --Q 4096 --I 200000 --scan 0) 2>&1 | grep -v amdgpu.ids > $out/fused_microbench.txt; cat $out/fused_microbench.txt
echo "== scan rows"; python3 scripts/scan_rows_probe.py 2>&1 | grep -v amdgpu.ids > $out/scan_rows.txt; cat $out/scan_rows.txt
echo "== ivf"; python3 scripts/ivf_probe.py 2>&1 | grep -v amdgpu.ids > $out/ivf_probe.txt; cat $out/ivf_probe.txt
echo "== index build"; python3 scripts/index_build_probe.py 2>&1 | grep -v amdgpu.ids > $out/index_build_probe.txt; cat $out/index_build_probe.txt
echo "== static checks + placement sweep"
(python3 scripts/check_mfma_hazards.py; python3 scripts/check_lds_hazards.py; bash scripts/placement_sweep.sh test 7) > $out/static_checks_and_placement_sweep.txt 2>&1; tail -20 $out/static_checks_and_placement_sweep.txt
fi
find $out -name "*.csv" -size +8M -delete
