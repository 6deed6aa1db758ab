mkdir -p gpurun_out
timeout 900 python tools/ab_stage.py tools/abl/libfa_tm0.so tools/abl/libfa_tm.so --stages fwd,dq,dkdv --rounds 7 --iters 5 --only "fp16 d128 512 causal,fp16 d128 1k causal,fp16 d128 2k causal,fp16 d128 4k causal,fp16 d128 3k causal,fp16 d64 2k causal,gqa 4k causal,bf16 d128 8k causal" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s4_ab_tile_major.log
