mkdir -p gpurun_out/r3b
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x -k "64 or d64 or golden or oracle or fuzz or dense or varlen" 2>&1 | tail -4
python tools/ab_stage.py tools/abl/libfa_d64bn64.so tools/abl/libfa_d64bn128.so --stages fwd --only "d64" --rounds 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3b/ab_fwd_d64.log
