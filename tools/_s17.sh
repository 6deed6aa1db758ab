mkdir -p gpurun_out
timeout 900 python tools/ab_stage.py tools/abl/libfa_lkw0.so tools/abl/libfa_lkw.so --stages fwd --rounds 7 --iters 3 --only "c3 fp16,c5shard,c2 fp16,fp16 d128 2k,fp16 d128 4k causal,bf16 d128 8k causal,c4 bf16" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s17_ab_late_k_wait.log
timeout 600 python tools/soak.py --iters 300 --policy mfma16 2>&1 | tail -12
