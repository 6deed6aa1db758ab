import sys, torch
sys.path.insert(0, "flash-attention-turing_amd"); sys.path.insert(0, "tests")
import flash_attn_turing as F, _util as U
dev = torch.device("cuda:0")
gen = torch.Generator(device="cpu").manual_seed(1234)
for (sq, sk, causal, b, h, hk, d) in ((2, 2, True, 1, 4, 4, 128), (1, 2, True, 3, 4, 2, 128), (63, 63, False, 3, 2, 1, 128)):
    for trial in range(3):
        q = torch.randn(b, sq, h, d, generator=gen).to(dev, torch.float16)
        k = torch.randn(b, sk, hk, d, generator=gen).to(dev, torch.float16)
        v = torch.randn(b, sk, hk, d, generator=gen).to(dev, torch.float16)
        do = torch.randn(b, sq, h, d, generator=gen).to(dev, torch.float16)
        o, lse = F.fwd(q, k, v, causal)
        dq, dk, dv = F.bwd(q, k, v, o, lse, do, causal)
        x = U.torch_attention_ref(q, k, v, do, causal, device="cpu", dtype=torch.float64)
        for name, got, e in (("O", o, x[0]), ("dQ", dq, x[2]), ("dK", dk, x[3]), ("dV", dv, x[4])):
            z = (e == 0)
            if z.any():
                g = got.float().cpu()
                print(sq, sk, causal, name, "exact zeros", int(z.sum()), "max |x| there", g[z].abs().max().item(), "o==v row0:", torch.equal(o[:, 0], v[:, 0].repeat_interleave(h // hk, 1)) if causal else None,
                      "lse-s", (lse[:, :, 0].cpu().double() - (q[:, 0].double().cpu() * k[:, 0].double().cpu().repeat_interleave(h // hk, 1)).sum(-1) / d ** 0.5).abs().max().item() if causal else None)
