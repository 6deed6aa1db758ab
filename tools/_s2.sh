mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/s2_pytest.log
cat gpurun_out/s2_pytest.log
timeout 400 python tools/ab_policy_sweep.py --seqs 512,1024,2048,4096,8192,16384 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s2_policy_sweep.log
