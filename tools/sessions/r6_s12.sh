#!/bin/bash
# round 6, GPU session 12: phase stamps of the forward's steady loop (-DFA_FWD_TIMING build) and what the stamps themselves cost
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s12; mkdir -p $O
A=tools/abl
export PYTHONUNBUFFERED=1
timeout 300 python tools/phase_timing_fwd.py $A/libfa_ftim.so > $O/fwd16_phase_timing.log 2>&1
grep -v amdgpu.ids $O/fwd16_phase_timing.log
timeout 400 python tools/ab_stage.py $A/libfa_base.so $A/libfa_ftim.so --only "c3 fp16,c5shard,c4 bf16" --stages fwd --rounds 5 > $O/fwd_stamp_cost_ab.log 2>&1
grep -v amdgpu.ids $O/fwd_stamp_cost_ab.log
