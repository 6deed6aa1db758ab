python tools/ab_stage.py tools/abl/libfa_kv64pf3.so tools/abl/libfa_kv64pf2.so --stages dkdv --only "d64,c4 bf16" --rounds 7 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3b/ab_dkdv_d64_pf.log
