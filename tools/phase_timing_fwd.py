#!/usr/bin/env python3
"""Read the per-wave phase cycle counters a -DFA_FWD_TIMING build of fa_fwd_pp16.hip leaves in the LSE tensor (development aid; LSE is wrong in such a build).
Usage: phase_timing_fwd.py LIB.so      prints, per wave group, the mean cycles per steady-loop step of: matrix phase, barrier behind it, LDS-DMA requests,
                                        softmax pass, fragment prefetch + counted DMA wait, barrier behind that (s_memtime ticks = shader cycles)"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
from flash_attn_turing import capi
L = ctypes.CDLL(os.path.abspath(sys.argv[1]))
L.fa_run_mha_fwd.argtypes = [ctypes.POINTER(capi.FwdParams), ctypes.c_void_p]
if hasattr(L, "fa_set_kernel_policy"):
    L.fa_set_kernel_policy(1)        # the 16x16x32 set, whatever the size
names = ["matrix phase", "barrier 1", "dma requests", "softmax pass", "prefetch+vmcnt", "barrier 2"]
dev = torch.device("cuda:0")
for (b, s, h, d, causal, dt) in ((4, 16384, 32, 128, True, torch.float16), (4, 16384, 32, 128, False, torch.float16), (4, 8192, 32, 128, False, torch.bfloat16),
                                 (1, 4096, 8, 128, False, torch.float16), (4, 8192, 32, 64, False, torch.float16)):
    g = torch.Generator(device=dev).manual_seed(0)
    q, k, v = (torch.randn(b, s, h, d, device=dev, dtype=dt, generator=g) for _ in range(3))
    o = torch.empty_like(q)
    lse = torch.zeros(b, h, s, device=dev, dtype=torch.float32)
    st = torch.cuda.current_stream().cuda_stream
    pf = capi.fwd_params(q, k, v, o, lse, causal)
    for _ in range(3):
        assert L.fa_run_mha_fwd(ctypes.byref(pf), st) == 0
    torch.cuda.synchronize()
    t = lse.view(b * h * (s // 256), 8, 32)[:, :, :8].double().cpu()
    tag = f"b{b} s{s} h{h} d{d} {'fp16' if dt == torch.float16 else 'bf16'} causal={int(causal)}"
    for grp, gname in ((slice(0, 4), "waves 0-3 (group A)"), (slice(4, 8), "waves 4-7 (group B)")):
        tt = t[:, grp]
        n = tt[..., 6].sum().item()
        per = [tt[..., i].sum().item() / n for i in range(6)]
        print(f"{tag} {gname}: cycles/step  " + "  ".join(f"{nm} {x:7.1f}" for nm, x in zip(names, per)) + f"  total {sum(per):7.1f}")
    w = [t[:, i, :6].sum().item() / t[:, i, 6].sum().item() for i in range(8)]
    print(f"{tag} total per wave: " + " ".join(f"{x:7.1f}" for x in w) + "   (pure MFMA issue of a step: 68 x 16 = 1088 per wave, 2176 per SIMD and tile pair)")
    mw = [t[:, i, 0].sum().item() / t[:, i, 6].sum().item() for i in range(8)]
    sw = [t[:, i, 3].sum().item() / t[:, i, 6].sum().item() for i in range(8)]
    print(f"{tag} matrix phase per wave: " + " ".join(f"{x:7.1f}" for x in mw) + "   softmax pass per wave: " + " ".join(f"{x:7.1f}" for x in sw))
