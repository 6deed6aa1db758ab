timeout 600 python -m pytest tests/test_attention_gpu.py -m gpu -q -x -k "d64_forward_tile or golden or oracle" 2>&1 | tail -3
python tools/ab_fwd_seqs.py tools/abl/libfa_d64bn64.so tools/abl/libfa_d64bn128.so tools/abl/libfa_d64auto.so --d 64 --rounds 5 2>&1 | grep -v amdgpu.ids
