python tools/ab_stage.py tools/abl/libfa_kvn0.so tools/abl/libfa_kvn1.so --stages dkdv --only "c4 bf16,bf16 d128 8k causal,c3 fp16" --rounds 7 2>&1 | grep -v amdgpu.ids | cut -c1-150
