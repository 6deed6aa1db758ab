tools/powerbench > gpurun_out/powerbench.log 2>&1
{
echo "# FA_PP16_FOLD_MAX=1 (Q pre-scaled by log2(e)/sqrt(d) and rounded back to fp16 / bf16; score MFMA chains start from -running_max; P = exp2(score)):"
echo "# A = fa_fwd_pp_kernel (32x32x16), B = fa_fwd_pp16_kernel as shipped, C = fa_fwd_pp16_kernel with the fold; tools/ab_stage.py"
python tools/ab_stage.py tools/abl/libfa_m32.so tools/abl/libfa_m16.so tools/abl/libfa_fold.so --stages fwd --only "c3 fp16,c4 bf16,c2 fp16,fp16 d128 2k" --rounds 7 2>&1 | grep -v amdgpu.ids
echo "# outputs of C against A on small / ragged / large-magnitude inputs (tools/ab_check.py): fp16, then bf16"
python tools/ab_check.py tools/abl/libfa_m32.so tools/abl/libfa_fold.so 2>&1 | grep -v amdgpu.ids
python tools/ab_check.py tools/abl/libfa_m32.so tools/abl/libfa_fold.so --dtype bf16 2>&1 | grep -v amdgpu.ids
echo "# the same for B (the shipped 16x16x32 kernel) against A"
python tools/ab_check.py tools/abl/libfa_m32.so tools/abl/libfa_m16.so 2>&1 | grep -v amdgpu.ids
python tools/ab_check.py tools/abl/libfa_m32.so tools/abl/libfa_m16.so --dtype bf16 2>&1 | grep -v amdgpu.ids
} > gpurun_out/fwd_fold_ab.log 2>&1
