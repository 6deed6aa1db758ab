mkdir -p gpurun_out
timeout 300 python tools/causal_wg_cost.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s9_causal_wg_cost.log
timeout 900 python tools/ab_stage.py tools/abl/libfa_cur.so tools/abl/libfa_kvprio4.so tools/abl/libfa_kvprio5.so tools/abl/libfa_mprio.so --stages fwd,dkdv --rounds 5 --iters 3 --only "c4 bf16,bf16 d128 8k causal,fp16 d128 2k,c3 fp16" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s9_ab_prio3.log
