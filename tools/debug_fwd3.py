#!/usr/bin/env python3
"""Development aid: V[key, d] = id of d's 32-wide block (+ 8 x id of the key's 16-key slice within its tile in the second run), so O shows which V fragment each
accumulator tile actually consumed."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
from flash_attn_turing import capi
L = ctypes.CDLL(sys.argv[1])
L.fa_run_mha_fwd.argtypes = [ctypes.POINTER(capi.FwdParams), ctypes.c_void_p]
dev = torch.device("cuda:0")
sq, sk, d = 256, int(sys.argv[2]) if len(sys.argv) > 2 else 128, 128
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(1, sq, 1, d, device=dev, dtype=torch.float16, generator=g)
k = torch.randn(1, sk, 1, d, device=dev, dtype=torch.float16, generator=g) * 0.0       # uniform attention: every key weight 1/sk
dblk = (torch.arange(d, device=dev) // 32).float()
for mode in ("db", "ts", "tile"):
    v = torch.zeros(1, sk, 1, d, device=dev, dtype=torch.float16)
    key = torch.arange(sk, device=dev)
    if mode == "db": v[0, :, 0, :] = dblk[None, :]
    if mode == "ts": v[0, :, 0, :] = ((key % 64) // 16).float()[:, None]
    if mode == "tile": v[0, :, 0, :] = (key // 64).float()[:, None]
    o = torch.empty_like(q); lse = torch.empty(1, 1, sq, device=dev, dtype=torch.float32)
    p = capi.fwd_params(q, k, v, o, lse, False)
    assert L.fa_run_mha_fwd(ctypes.byref(p), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    print(f"mode {mode}: expected per d-block: " + ("0 1 2 3" if mode == "db" else f"{v.float().mean().item():.3f} everywhere"))
    for row in (8, 40, 72, 104, 200, 232):
        print(f"   row {row:3d} (wave {row // 64} q-block {(row // 32) % 2}): " + " ".join(f"{o[0, row, 0, 32 * b + 5].item():7.4f}" for b in range(4)))
