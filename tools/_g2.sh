mkdir -p gpurun_out/r3d gpurun_out/r3a
timeout 1800 python -m pytest tests -m gpu -q -rf --tb=short > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3a/pytest.log
python tools/sweep_efficiency.py > gpurun_out/r3d/sweep_efficiency.log 2>&1; echo "sweep rc=$?"
bash tools/round_evidence.sh > gpurun_out/r3d/evidence_run.log 2>&1; tail -3 gpurun_out/r3d/evidence_run.log
