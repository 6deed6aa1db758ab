timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/s45_pytest.log; cat gpurun_out/s45_pytest.log
timeout 900 python tools/soak.py --iters 150 > gpurun_out/s45_soak_auto.log 2>&1; tail -2 gpurun_out/s45_soak_auto.log
timeout 900 python tools/soak.py --iters 100 --policy mfma16 > gpurun_out/s45_soak_mfma16.log 2>&1; tail -2 gpurun_out/s45_soak_mfma16.log
