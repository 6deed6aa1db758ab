python tools/ab_fwd_seqs.py tools/abl/libfa_fwd0.so tools/abl/libfa_fwdearly.so --d 128 --rounds 7 2>&1 | grep -v amdgpu.ids
python tools/ab_stage.py tools/abl/libfa_fwd0.so tools/abl/libfa_fwdearly.so --stages fwd --only "c3 fp16,fp16 d128 1k,d128 512,d64 8k" --rounds 5 2>&1 | grep -v amdgpu.ids
