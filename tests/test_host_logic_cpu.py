"""CPU: host-side logic of the drop-in module (no kernels run)."""
import pytest
import torch

import flash_attn_turing as F


def test_module_surface_matches_reference_exports():
    # reference csrc/flash_attn/flash_api.cpp:471-476 + README's flash_attn_func
    for name in ("fwd", "bwd", "varlen_fwd", "varlen_bwd", "flash_attn_func"):
        assert callable(getattr(F, name))
    assert F.abi_version() == 4
    # round 4: the differentiable wrappers run C++ autograd nodes of the host module; CPU tensors fail in the same loud way through them
    assert callable(F._C.attn_autograd) and callable(F._C.attn_varlen_autograd)
    x = torch.zeros(1, 8, 2, 128, dtype=torch.float16, requires_grad=True)
    with pytest.raises(RuntimeError, match="GPU"):
        F.flash_attn_func(x, x, x, causal=True)
    with pytest.raises(RuntimeError, match="rank-3"):
        F.flash_attn_varlen_func(x, x, x, torch.zeros(2, dtype=torch.int32), torch.zeros(2, dtype=torch.int32), 8, 8)


def test_cpu_tensors_are_rejected_not_silently_computed():
    x = torch.zeros(1, 8, 2, 128, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="GPU"):
        F.fwd(x, x, x, False)
    with pytest.raises(RuntimeError, match="rank-4"):
        F.fwd(x[0], x, x, False)
    with pytest.raises(RuntimeError, match="rank-3"):
        F.varlen_fwd(x, x, x, torch.zeros(2, dtype=torch.int32), torch.zeros(2, dtype=torch.int32), 8, 8, False)


@pytest.mark.parametrize("batch,h,hk,world", [(4, 32, 32, 1), (4, 32, 32, 2), (4, 32, 32, 4), (4, 32, 32, 8),
                                                (32, 32, 32, 8), (3, 6, 3, 2), (1, 8, 2, 8), (5, 4, 4, 3)])
def test_shard_plans_partition_the_problem(batch, h, hk, world):
    plans = F.plan_shards(batch, h, hk, world)
    assert len(plans) == world
    seen = set()
    for p in plans:
        assert (p.head_stop - p.head_start) == (p.head_k_stop - p.head_k_start) * (h // hk)   # GQA groups stay whole
        for b in range(p.batch_start, p.batch_stop):
            for g in range(p.head_k_start, p.head_k_stop):
                assert (b, g) not in seen
                seen.add((b, g))
    assert len(seen) == batch * hk                                 # every (batch, kv head) exactly once
    assert sum(p.n_units for p in plans) == batch * h


def test_shard_views_are_strided_slices():
    q = torch.zeros(4, 16, 8, 64)
    k = torch.zeros(4, 16, 2, 64)
    p = F.plan_shards(4, 8, 2, 8)[5]
    qs, ks = F.shard_tensor(q, p, False), F.shard_tensor(k, p, True)
    assert qs.shape == (1, 16, 4, 64) and ks.shape == (1, 16, 1, 64)
    assert qs.data_ptr() == q[p.batch_start, 0, p.head_start].data_ptr()


def test_flash_attn_func_legacy_signature_validation():
    x = torch.zeros(1, 8, 2, 128, dtype=torch.float16)
    with pytest.raises(ValueError):
        F.flash_attn_func(x, x, x, 1, 8, 2, 64)
    with pytest.raises(TypeError):
        F.flash_attn_func(x, x, x, 1, 2)
