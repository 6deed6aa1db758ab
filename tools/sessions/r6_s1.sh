#!/bin/bash
# round 6, GPU session 1: the right-shape ceiling probes (clockbench 16x16x32 mixes + the ping-pong structure probe, chip-wide and on one CU), the MFMA cadence
# microbenchmarks, the small-grid policy sweep on the round-5 thresholds, LDS-DMA request staggering (forward / dK/dV), dK/dV16 phase stamps + per-workgroup cost
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s1; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 300 tools/clockbench > $O/clockbench.log 2>&1
timeout 120 tools/clockbench --grid 1 --rows "16x16x32" --reps 3 > $O/clockbench_one_cu.log 2>&1
timeout 120 tools/ubench > $O/ubench.log 2>&1
cat $O/clockbench.log $O/clockbench_one_cu.log; sed -n '/round 6/,$p' $O/ubench.log
timeout 300 python tools/check_variant_fwd.py --lib $A/libfa_fstag.so > $O/check_fstag.log 2>&1; tail -n 3 $O/check_fstag.log
timeout 500 python tools/ab_stage.py $A/libfa_base.so $A/libfa_fstag.so --only "c3 fp16,c5shard,c2 fp16,fp16 d128 4k causal,c4 bf16,fp16 d128 2k" --stages fwd --rounds 7 > $O/fwd_dma_stagger_ab.log 2>&1
grep -v amdgpu.ids $O/fwd_dma_stagger_ab.log
timeout 500 python tools/ab_stage.py $A/libfa_base.so $A/libfa_kvstag1.so $A/libfa_kvstag2.so --only "c4 bf16,bf16 d128 8k causal,c2 fp16,fp16 d128 2k,fp16 d128 4k causal,c3 fp16" --stages dkdv --rounds 7 > $O/dkdv_dma_stagger_ab.log 2>&1
grep -v amdgpu.ids $O/dkdv_dma_stagger_ab.log
timeout 300 python tools/phase_timing_dkdv.py $A/libfa_kvtim.so --layout 16 > $O/dkdv16_phase_timing.log 2>&1
grep -v amdgpu.ids $O/dkdv16_phase_timing.log
timeout 400 python tools/ab_stage.py $A/libfa_base.so --only "c4 bf16,bf16 d128 8k causal" --stages dq,dkdv --rounds 5 > $O/bwd_stage_times.log 2>&1
grep -v amdgpu.ids $O/bwd_stage_times.log
for g in "1 8" "1 32" "2 16"; do set -- $g; timeout 400 python tools/ab_policy_sweep.py --b $1 --h $2 --d 128 --dtype fp16 --seqs 512,1024,2048,4096,8192 >> $O/policy_small_grids.log 2>&1; done
grep -v amdgpu.ids $O/policy_small_grids.log
