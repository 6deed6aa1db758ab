mkdir -p gpurun_out/r3b
python tools/ab_stage.py tools/abl/libfa_lock2.so tools/abl/libfa_lock4.so --stages dkdv --only "c4 bf16,8k causal,c2 fp16,fp16 d128 1k,mqa causal,gqa 4k,d128 512" --rounds 5 > gpurun_out/r3b/ab_dkdv5.log 2>&1
cat gpurun_out/r3b/ab_dkdv5.log
python tools/phase_timing_dkdv.py tools/abl/libfa_lock4time.so > gpurun_out/r3b/kvtime_lock4.log 2>&1
cat gpurun_out/r3b/kvtime_lock4.log
