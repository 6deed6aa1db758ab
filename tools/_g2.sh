mkdir -p gpurun_out/r3b
python tools/ab_stage.py tools/abl/libfa_fwd0.so tools/abl/libfa_fwdprio1.so tools/abl/libfa_fwdprio2.so --stages fwd --only "c3 fp16,c4 bf16,c2 fp16,fp16 d128 1k,d128 512,d64 8k" --rounds 7 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3b/ab_fwd_prio.log
