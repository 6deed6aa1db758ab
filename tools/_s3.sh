mkdir -p gpurun_out
timeout 600 python tools/ab_stage.py tools/abl/libfa_ml1.so tools/abl/libfa_w4.so --stages fwd --rounds 5 --iters 3 --only "c5shard,c2,fp16 d128 2k,fp16 d128 1k,fp16 d128 512,c4 bf16" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s3_ab_w4.log
timeout 900 python -m pytest tests/test_kernel_sets_gpu.py -q -k "mfma16 and (reference_grid or varlen_random or rising or policy_switches)" 2>&1 | grep -E "^E  |^FAILED|passed|failed|AssertionError" | cut -c1-400 | head -150 > gpurun_out/s3_pytest.log
cat gpurun_out/s3_pytest.log
