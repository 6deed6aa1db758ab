"""flash_attn_turing — MI355X (gfx950 / CDNA4) drop-in for ssiu/flash-attention-turing.

Same import surface as the reference's compiled module (reference
csrc/flash_attn/flash_api.cpp:471-476; used as `from flash_attn_turing import fwd, bwd,
varlen_fwd, varlen_bwd`, reference test_flash_attn.py:12-17) plus the `flash_attn_func`
convenience wrapper the reference README documents (README.md:28-48).

The hot path is hand-written HIP in ``csrc/libflash_attn_gfx950.so`` behind the C ABI of
``include/flash_attn_gfx950.h``.  There is NO fallback: if the compiled pieces are missing the
import fails loudly (build them with ``python flash-attention-turing_amd/build.py``).
"""
import os as _os

_HERE = _os.path.dirname(_os.path.abspath(__file__))
_CSRC = _os.path.join(_os.path.dirname(_HERE), "csrc")
# in-tree build: ../csrc/libflash_attn_gfx950.so; installed (`pip install .`, setup.py): the library ships inside the package
LIBRARY_PATH = next((c for c in (_os.path.join(_HERE, "libflash_attn_gfx950.so"), _os.path.join(_CSRC, "libflash_attn_gfx950.so")) if _os.path.exists(c)),
                    _os.path.join(_CSRC, "libflash_attn_gfx950.so"))
EXTENSION_PATH = _os.path.join(_HERE, "_C.so")

if not _os.path.exists(LIBRARY_PATH) or not _os.path.exists(EXTENSION_PATH):
    raise ImportError(
        "flash_attn_turing: compiled HIP extension not found "
        f"({LIBRARY_PATH if not _os.path.exists(LIBRARY_PATH) else EXTENSION_PATH}). "
        "Run `python flash-attention-turing_amd/build.py` or `pip install .` (needs hipcc); there is no CPU/PyTorch fallback."
    )

import torch as _torch  # noqa: E402,F401  (libtorch must be loaded before _C)

from . import _C  # noqa: E402
from ._C import fwd, bwd, varlen_fwd, varlen_bwd  # noqa: E402,F401
from .interface import (  # noqa: E402,F401
    flash_attn_func,
    flash_attn_varlen_func,
    FlashAttnFunc,
    FlashAttnVarlenFunc,
)
from .sharding import ShardPlan, plan_shards, shard_tensor, problem_policy  # noqa: E402,F401

__all__ = [
    "fwd", "bwd", "varlen_fwd", "varlen_bwd",
    "flash_attn_func", "flash_attn_varlen_func", "FlashAttnFunc", "FlashAttnVarlenFunc",
    "ShardPlan", "plan_shards", "shard_tensor", "problem_policy", "LIBRARY_PATH", "EXTENSION_PATH",
    "set_kernel_policy", "kernel_name",
]
__version__ = "0.1.0"


def set_kernel_policy(policy) -> str:
    """Which of the two head_dim-128 kernel sets serves the launches of this process: "auto" (default: per launch by sequence length, mask and
    how far the launch fills the chip; `problem_policy` gives the shards of a problem the whole problem's choice), "mfma32" or "mfma16" (C ABI fa_set_kernel_policy; both sets meet the same tolerances, they differ in speed only).  Returns the
    previous policy's name.  No counterpart in the reference."""
    from . import capi

    names = {"mfma32": capi.POLICY_MFMA32, "mfma16": capi.POLICY_MFMA16, "auto": capi.POLICY_AUTO}
    if policy not in names:
        raise ValueError(f"policy must be one of {sorted(names)}")
    prev = capi.set_kernel_policy(names[policy])
    return next(k for k, v in names.items() if v == prev)


def kernel_name(stage, batch, seqlen_q, seqlen_k, nheads, head_dim, causal, dtype=None) -> str:
    """the kernel a launch of this shape goes to under the current policy; stage in {"fwd", "dq", "dkdv"}; dtype torch.float16 (default) /
    torch.bfloat16 or "fp16" / "bf16" (C ABI fa_kernel_name_dtype)"""
    from . import capi

    name = {None: "fp16", _torch.float16: "fp16", _torch.bfloat16: "bf16", "fp16": "fp16", "bf16": "bf16"}.get(dtype)
    if name is None:
        raise ValueError("dtype must be torch.float16 or torch.bfloat16")
    return capi.kernel_name(stage, batch, seqlen_q, seqlen_k, nheads, head_dim, causal, name)


def abi_version() -> int:
    return _C.abi_version()


def build_info() -> str:
    return _C.build_info()
