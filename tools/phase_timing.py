#!/usr/bin/env python3
"""Read the per-wave phase cycle counters a -DFA_X_TIMING build of the forward leaves in the LSE rows (development aid)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
from flash_attn_turing import capi
L = ctypes.CDLL(sys.argv[1])
L.fa_run_mha_fwd.argtypes = [ctypes.POINTER(capi.FwdParams), ctypes.c_void_p]
dev = torch.device("cuda:0")
for (b, s, h, d, causal) in ((4, 8192, 32, 128, False), (4, 16384, 32, 128, True), (4, 2048, 32, 128, False)):
    g = torch.Generator(device=dev).manual_seed(0)
    q, k, v = (torch.randn(b, s, h, d, device=dev, dtype=torch.float16, generator=g) for _ in range(3))
    o = torch.empty_like(q); lse = torch.zeros(b, h, s, device=dev, dtype=torch.float32)
    p = capi.fwd_params(q, k, v, o, lse, causal)
    for _ in range(3):
        assert L.fa_run_mha_fwd(ctypes.byref(p), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    t = lse.view(b, h, s // 32, 32)[..., 8:14].double()          # (b, h, waves, 6)
    t = t.view(b, h, s // 256, 8, 6)
    n = t[..., 5]
    ok = n > 0
    for grp, name in ((slice(0, 4), "group A (waves 0-3)"), (slice(4, 8), "group B (waves 4-7)")):
        tt = t[:, :, :, grp]; nn = tt[..., 5]; m = nn > 0
        per = [(tt[..., i][m] / nn[m]).mean().item() for i in range(5)]
        print(f"b{b} s{s} causal={causal} {name}: cycles/iter  PV {per[0]:7.1f}  QK {per[1]:7.1f}  dmaK+barrier {per[2]:7.1f}  dmaV+softmax {per[3]:7.1f}  vmcnt+barrier {per[4]:7.1f}  total {sum(per):7.1f}  (ideal MFMA 2x512)")
