mkdir -p gpurun_out
timeout 900 python tools/ab_stage.py tools/abl/libfa_cur.so tools/abl/libfa_cw1.so --stages fwd --rounds 7 --iters 3 --only "c3 fp16,c5shard,c2 fp16,fp16 d128 2k,fp16 d128 4k causal,bf16 d128 8k causal,c4 bf16,fp16 d128 4k causal b8" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s16_ab_counted_wait.log
