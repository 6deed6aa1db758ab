mkdir -p gpurun_out/r3a
timeout 1800 python -m pytest tests -m gpu -q -rf --tb=short > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r3a/pytest.log
grep -E "^E  " gpurun_out/r3a/pytest.log | cut -c1-400 | head -12
