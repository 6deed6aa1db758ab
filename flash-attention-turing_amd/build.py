#!/usr/bin/env python3
"""In-tree build of the MI355X attention library and its PyTorch host module.

Replaces the reference's CUDA `setup.py` (reference setup.py:20-49: CUDAExtension with nvcc
-arch=sm_75).  Two artefacts, both written next to the sources so they travel with the tree:

  csrc/libflash_attn_gfx950.so       HIP kernels + C ABI (include/flash_attn_gfx950.h),
                                     built with `hipcc --offload-arch=gfx950`; no torch dependency.
  flash_attn_turing/_C.so            host module (pybind11 + torch C++ API, plain g++ — no HIP
                                     sources, so torch's hipify pass is never involved); it
                                     links against the library above and exports
                                     fwd / bwd / varlen_fwd / varlen_bwd exactly like
                                     reference csrc/flash_attn/flash_api.cpp:471-476.

hipcc cross-compiles for gfx950 without a GPU.  Usage: `python build.py [--force] [--no-torch]`.
"""
import argparse
import hashlib
import os
import subprocess
import sys
import sysconfig
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
PKG = os.path.join(HERE, "flash_attn_turing")
INCLUDE = os.path.join(ROOT, "include")

LIB_NAME = "libflash_attn_gfx950.so"
LIB_PATH = os.path.join(CSRC, LIB_NAME)
EXT_PATH = os.path.join(PKG, "_C.so")

HIP_SOURCES = ["fa_fwd_pp.hip", "fa_fwd_pp16.hip", "fa_bwd.hip", "fa_bwd_dq16.hip", "fa_bwd_dkdv16.hip", "fa_capi.hip"]
# per-file extra flags.  fa_fwd_pp16.hip: hipcc's SLP vectoriser packs the softmax row-sum adds and the O rescale into v_pk_* on register
# pairs it first has to assemble from the 4-register MFMA tiles: ~200 v_mov_b64 per three tiles and 44-116 bytes of spills on the hot path
EXTRA_FLAGS = {"fa_fwd_pp16.hip": ["-fno-slp-vectorize"], "fa_bwd_dq16.hip": ["-fno-slp-vectorize"], "fa_bwd_dkdv16.hip": ["-fno-slp-vectorize"]}
HIP_HEADERS = ["fa_device.hpp", "fa_params.hpp", "fa_bwd_dkdv_common.hpp", os.path.join(INCLUDE, "flash_attn_gfx950.h")]
# -amdgpu-mfma-vgpr-form: builtin MFMAs keep their result in VGPRs even in kernels that may use the
# accumulator half of the register file (the dK/dV kernel parks its 128 long-lived accumulator
# registers in AGPRs through LP<T>::mfma_agpr); without it hipcc selects the AGPR form for every MFMA
# of such a kernel and shuttles the softmax operands with v_accvgpr_read/write.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form",
               "-DNDEBUG", "-Wall", "-Wno-unused-function"]


def _run(cmd, **kw):
    print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, **kw)


def _digest(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale(target, stamp_value):
    stamp = target + ".stamp"
    if not os.path.exists(target) or not os.path.exists(stamp):
        return True
    with open(stamp) as f:
        return f.read().strip() != stamp_value


def _write_stamp(target, stamp_value):
    with open(target + ".stamp", "w") as f:
        f.write(stamp_value)


# Debug builds of the library for tests/test_dma_protocol_gpu.py (round 5): the LDS-DMA protocols of fa_fwd_pp16 and fa_bwd_dkdv16 under adversarial
# timing.  Only the two sources that carry the switches are recompiled; everything else is the product's objects.  The product's ISA does not change.
DEBUG_SOURCES = ["fa_fwd_pp16.hip", "fa_bwd_dkdv16.hip"]
DEBUG_VARIANTS = {
    "dma_late": ["-DFA_PP16_DMA_DEBUG=1", "-DFA_KV16_DMA_DEBUG=1"],          # every request issued directly in front of the wait that retires it
    "dma_sleepy": ["-DFA_PP16_DMA_DEBUG=2", "-DFA_KV16_DMA_DEBUG=2"],        # group B / waves 4-7 sleep ~4000 cycles in front of every request
    "dma_racy": ["-DFA_PP16_ROLE_DMA=0", "-DFA_PP16_RACY=1"],                # the documented WRONG counted-wait form of round 4, normal timing
    "dma_racy_late": ["-DFA_PP16_ROLE_DMA=0", "-DFA_PP16_RACY=1", "-DFA_PP16_DMA_DEBUG=1"],      # ... under late issue: must FAIL
}
DEBUG_DIR = os.path.join(CSRC, "debug")


def debug_library_path(name):
    return os.path.join(DEBUG_DIR, f"libfa_{name}.so")


M0_GUARD_SOURCES = ["fa_fwd_pp.hip", "fa_fwd_pp16.hip", "fa_bwd.hip", "fa_bwd_dq16.hip", "fa_bwd_dkdv16.hip"]


def m0_uses_outside_asm(asm_text):
    """mentions of the M0 register in gfx950 assembly outside `;;#ASMSTART .. ;;#ASMEND` blocks (= in code hipcc generated itself)"""
    import re

    outside = re.sub(r";;#ASMSTART.*?;;#ASMEND", "", asm_text, flags=re.S)
    return len(re.findall(r"\bm0\b", outside))


def hipcc_path():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def build_kernels(force=False):
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HIP_HEADERS]
    stamp = _digest(srcs + hdrs, " ".join(HIPCC_FLAGS) + repr(sorted(EXTRA_FLAGS.items())))
    # the same digest goes INTO the library (fa_build_info()): a profile summary records it, bench.py compares it with the library
    # it is timing, so a committed rocprof table can be tied to the kernels that produced it
    digest_flag = f'-DFA_SOURCE_DIGEST="{stamp[:12]}"'
    if not force and not _stale(LIB_PATH, stamp):
        print(f"[build] {LIB_NAME} up to date")
        return LIB_PATH
    t0 = time.time()
    objs = []
    procs = []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        cmd = [hipcc_path()] + HIPCC_FLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-I", CSRC, "-I", INCLUDE, "-c", s, "-o", o]
        if os.path.basename(s) == "fa_capi.hip":
            cmd.insert(1, digest_flag)
        print("[build]", " ".join(cmd), flush=True)
        procs.append(subprocess.Popen(cmd))
    # M0 guard: every kernel issues its LDS-DMA from inline asm WITHOUT saving / restoring M0, which is only sound while nothing hipcc itself
    # generates for those kernels touches M0 (no compiler-visible LDS-DMA, no indirect register indexing, no readlane-by-M0).  A different compiler or
    # a future edit could change that silently, so the build looks: the device assembly of every kernel source, outside the hand-written
    # `;;#ASMSTART .. ;;#ASMEND` statements, must not mention m0 (the same scan as tests/test_kernel_resources_cpu.py, here so that a wheel or an
    # in-tree build made without running the tests cannot ship it).
    scans = []
    for s in srcs:
        if os.path.basename(s) in M0_GUARD_SOURCES:
            asm = s[:-4] + ".guard.s"
            cmd = [hipcc_path()] + HIPCC_FLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-I", CSRC, "-I", INCLUDE, "-Wno-unused-command-line-argument", "--cuda-device-only", "-S", s, "-o", asm]
            scans.append((s, asm, subprocess.Popen(cmd)))
    sys.path.insert(0, HERE)
    import mfma_hazards

    try:
        for p in procs:
            if p.wait() != 0:
                raise RuntimeError("hipcc failed")
        for s, asm, p in scans:
            if p.wait() != 0:
                raise RuntimeError("hipcc failed (M0 guard pass)")
            with open(asm) as f:
                n = m0_uses_outside_asm(f.read())
            # MFMA-result guard (round 5): the accumulating MFMAs of the 16x16x32 kernels are inline asm, hipcc's hazard recogniser does not see them, and the forward's
            # steady loop relies on the instructions between a tile's last MFMA and the first score read instead of a pad (fa_fwd_pp16.hip: `covered`).  mfma_hazards.py
            # walks the ISA; a finding means this compiler scheduled a reader too close behind its MFMA.
            found = {k: v for k, v in mfma_hazards.scan_file(asm).items() if v}
            if found:
                k, v = next(iter(found.items()))
                raise RuntimeError(f"{os.path.basename(s)}: {sum(len(x) for x in found.values())} instruction(s) touch an MFMA result before it has landed, e.g. in {k}: "
                                   f"line {v[0][0]} `{v[0][1]}` {v[0][4]} of {v[0][5]} wait states behind `{v[0][3]}` (flash-attention-turing_amd/mfma_hazards.py)")
            if n:
                raise RuntimeError(f"{os.path.basename(s)}: {n} use(s) of M0 outside the hand-written asm statements - the unsaved-M0 LDS-DMA of these kernels is "
                                   "no longer safe with this compiler / this edit (fa_params.hpp FA_BWD_DMA_SAVE_M0, fa_device.hpp dma16_to_lds_hidden)")
    finally:
        # (whatever failed, on whichever file - no stray processes, no .guard.s left in the source directory)
        for p in procs + [x[2] for x in scans]:
            if p.poll() is None:
                p.wait()
        for _, asm, _ in scans:
            if os.path.exists(asm):
                os.remove(asm)
    _run([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs)
    _write_stamp(LIB_PATH, stamp)
    print(f"[build] {LIB_NAME} built in {time.time() - t0:.1f}s")
    return LIB_PATH


def build_debug_variants(force=False):
    """csrc/debug/libfa_<variant>.so: the product library with DEBUG_SOURCES rebuilt under a variant's switches (needs build_kernels() first)"""
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HIP_HEADERS]
    base = _digest(srcs + hdrs, " ".join(HIPCC_FLAGS) + repr(sorted(EXTRA_FLAGS.items())))
    os.makedirs(DEBUG_DIR, exist_ok=True)
    todo = [(n, f) for n, f in DEBUG_VARIANTS.items() if force or _stale(debug_library_path(n), base + repr(f))]
    if not todo:
        print("[build] debug variants up to date")
        return
    # the variants link the PRODUCT's objects of every source they do not rebuild: an up-to-date library whose objects were cleaned (or never shipped) is rebuilt first
    missing = [s for s in HIP_SOURCES if s not in DEBUG_SOURCES and not os.path.exists(os.path.join(CSRC, s[:-4] + ".o"))]
    if missing:
        print(f"[build] product objects missing ({', '.join(missing)}): rebuilding the library before the debug variants")
        build_kernels(force=True)
    t0 = time.time()
    procs = []
    for n, flags in todo:
        for s in DEBUG_SOURCES:
            o = os.path.join(DEBUG_DIR, f"{n}_{s[:-4]}.o")
            cmd = [hipcc_path()] + HIPCC_FLAGS + EXTRA_FLAGS.get(s, []) + flags + ["-I", CSRC, "-I", INCLUDE, "-c", os.path.join(CSRC, s), "-o", o]
            procs.append(subprocess.Popen(cmd))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed (debug variant)")
    for n, flags in todo:
        objs = [os.path.join(DEBUG_DIR, f"{n}_{s[:-4]}.o") if s in DEBUG_SOURCES else os.path.join(CSRC, s[:-4] + ".o") for s in HIP_SOURCES]
        _run([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", debug_library_path(n)] + objs)
        _write_stamp(debug_library_path(n), base + repr(flags))
        for s in DEBUG_SOURCES:
            os.remove(os.path.join(DEBUG_DIR, f"{n}_{s[:-4]}.o"))
    print(f"[build] {len(todo)} debug variant(s) built in {time.time() - t0:.1f}s")


def build_torch_module(force=False):
    import torch  # noqa: F401  (needed for include/library paths)
    from torch.utils import cpp_extension as ce

    src = os.path.join(CSRC, "flash_api.cpp")
    stamp = _digest([src, os.path.join(INCLUDE, "flash_attn_gfx950.h")], torch.__version__ + " rpath:$ORIGIN,$ORIGIN/../csrc")
    if not force and not _stale(EXT_PATH, stamp):
        print("[build] flash_attn_turing/_C.so up to date")
        return EXT_PATH
    t0 = time.time()
    tlib = ce.library_paths()[0]
    rocm = os.environ.get("ROCM_HOME", "/opt/rocm")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
           "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM", "-DTORCH_EXTENSION_NAME=_C", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    for inc in ce.include_paths():
        cmd += ["-isystem", inc]
    cmd += ["-isystem", os.path.join(rocm, "include"), "-I", INCLUDE, "-I", sysconfig.get_paths()["include"]]
    cmd += [src, "-o", EXT_PATH,
            "-L", tlib, "-lc10", "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10_hip", "-ltorch_hip",
            "-L", CSRC, "-l:" + LIB_NAME,
            "-L", os.path.join(rocm, "lib"), "-lamdhip64",
            "-Wl,-rpath," + tlib, "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/../csrc", "-Wl,-rpath," + os.path.join(rocm, "lib")]      # $ORIGIN: installed layout (setup.py)
    _run(cmd)
    _write_stamp(EXT_PATH, stamp)
    print(f"[build] flash_attn_turing/_C.so built in {time.time() - t0:.1f}s")
    return EXT_PATH


def build_all(force=False, torch_module=True, debug_variants=True):
    """debug_variants: the four test builds of tests/test_dma_protocol_gpu.py (csrc/debug/, ~8 extra compiles; never packaged).  An in-tree developer / CI build wants
    them (the GPU suite loads them), an installation does not: setup.py and `build.py --no-debug` leave them out."""
    build_kernels(force)
    if debug_variants:
        build_debug_variants(force)
    if torch_module:
        build_torch_module(force)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--no-torch", action="store_true", help="only build the C-ABI kernel library")
    ap.add_argument("--no-debug", action="store_true", help="skip the test builds of csrc/debug/ (tests/test_dma_protocol_gpu.py needs them)")
    a = ap.parse_args()
    build_all(a.force, not a.no_torch, not a.no_debug)
    sys.exit(0)
