#!/usr/bin/env python3
"""Build a named variant of the kernel library for A/B timing: tools/abl/libfa_<name>.so from the product sources plus -D switches.
Usage: build_variant.py name [-DFOO=1 ...]   (development aid; the product build is flash-attention-turing_amd/build.py)
FA_VARIANT_CSRC=dir builds from a scratch copy of csrc/ instead (an experiment on the sources while the product tree must stay as it is)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention-turing_amd"))
import build as b  # noqa: E402


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    csrc = os.environ.get("FA_VARIANT_CSRC", b.CSRC)
    out_dir = os.path.join(ROOT, "tools", "abl")
    os.makedirs(out_dir, exist_ok=True)
    objs, procs = [], []
    for src in b.HIP_SOURCES:
        o = os.path.join(out_dir, f"{name}_{src[:-4]}.o")
        objs.append(o)
        procs.append(subprocess.Popen([b.hipcc_path()] + b.HIPCC_FLAGS + b.EXTRA_FLAGS.get(src, []) + flags + ["-I", csrc, "-I", b.INCLUDE, "-c", os.path.join(csrc, src), "-o", o]))
    for p in procs:
        if p.wait() != 0:
            raise SystemExit("hipcc failed")
    lib = os.path.join(out_dir, f"libfa_{name}.so")
    subprocess.check_call([b.hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    for o in objs:
        os.remove(o)
    print(lib)


if __name__ == "__main__":
    main()
