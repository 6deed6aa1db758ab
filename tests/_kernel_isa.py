"""Compile a .hip file of the package to gfx950 assembly (hipcc cross-compiles without a GPU) and report, per kernel,
the compiler's own resource usage (-Rpass-analysis=kernel-resource-usage) plus what sits INSIDE its MFMA loops:
scratch (spill) traffic and v_accvgpr_* register shuffles.  Test infrastructure only."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "flash-attention-turing_amd")
sys.path.insert(0, PKG)
import build as _build  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _mfma_hazards  # noqa: E402

_KEYS = {"VGPRs": "vgprs", "AGPRs": "agprs", r"ScratchSize \[bytes/lane\]": "scratch_bytes", r"Occupancy \[waves/SIMD\]": "occupancy",
         r"LDS Size \[bytes/block\]": "lds_bytes"}


def _loops(body):
    """(label, text) of every loop, from the compiler's own block annotations: the block it marks `; =>This [Inner] Loop Header` plus
    every block marked `in Loop: Header=<that label>` (labelled blocks and fall-through `; %bb.N:` blocks alike), wherever the block
    placement put them - a rotated loop's body sits BEFORE its header in the text, and a label followed somewhere by a branch to it
    may just be a forward branch from a block placed later (that once made a kernel's accumulator zero-initialisation look like 128
    accumulator moves inside a loop)."""
    starts = [(m.start(), m.group(0)) for m in re.finditer(r"^(?:\.LBB\d+_\d+:|; %bb\.\d+:)[^\n]*$", body, re.M)]
    blocks = [(line, body[a:(starts[i + 1][0] if i + 1 < len(starts) else len(body))]) for i, (a, line) in enumerate(starts)]
    out = []
    for line, _ in blocks:
        m = re.match(r"\.(LBB\d+_\d+):.*Loop Header", line)
        if not m:
            continue
        hdr = m.group(1)[1:]                                   # "BB9_30"
        text = "".join(t for l, t in blocks if l.startswith("." + "L" + hdr + ":") or re.search(r"in Loop: Header=" + hdr + r"\b", l))
        out.append(("." + "L" + hdr, text))
    return out


def analyse(hip_source, extra_flags=()):
    src = os.path.join(_build.CSRC, hip_source)
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        cmd = [_build.hipcc_path()] + list(_build.HIPCC_FLAGS) + list(_build.EXTRA_FLAGS.get(hip_source, [])) + list(extra_flags) + ["-I", _build.CSRC, "-I", _build.INCLUDE, "--cuda-device-only", "-S", src,
                                                                                      "-o", asm, "-Rpass-analysis=kernel-resource-usage"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        txt = open(asm).read()
    kernels, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for pat, key in _KEYS.items():
            m = re.search(r"\s" + pat + r": (\d+)", line)
            if m:
                cur[key] = int(m.group(1))
    for name, info in kernels.items():
        if name + ":" not in txt:
            continue
        body = txt[txt.index(name + ":"):]
        # the function's extent is its .Lfunc_end label (an early `return` compiles to an early s_endpgm: cutting at the first
        # s_endpgm once truncated every dK/dV kernel to its prologue and made the loop checks pass vacuously)
        m = re.search(r"^\.Lfunc_end\d+:", body, re.M)
        body = body[:m.start()] if m else body[:body.rindex("s_endpgm")]
        info["mfma_total"] = len(re.findall(r"v_mfma", body))
        # results of MFMAs touched before they have landed (tests/_mfma_hazards.py: the asm-issued MFMAs are invisible to hipcc's hazard recogniser)
        bl = body.split("\n")
        info["mfma_hazards"] = _mfma_hazards.scan_kernel(bl, 0, len(bl))
        # M0 outside hand-written asm statements (;;#ASMSTART .. ;;#ASMEND): kernels whose LDS-DMA does not save / restore M0 rely on
        # hipcc itself never using it
        outside = re.sub(r";;#ASMSTART.*?;;#ASMEND", "", body, flags=re.S)
        info["m0_outside_asm"] = len(re.findall(r"\bm0\b", outside))
        info["loops"] = []
        for lab, seg in _loops(body):
            n = len(re.findall(r"v_mfma", seg))
            if n >= 8:
                info["loops"].append({"label": lab, "mfma": n, "scratch_ops": len(re.findall(r"scratch_(?:load|store)", seg)),
                                      "accvgpr_moves": len(re.findall(r"v_accvgpr_(?:read|write|mov)", seg)),
                                      "valu": len([1 for x in re.findall(r"^\s+(v_[a-z0-9_]+)", seg, re.M) if "mfma" not in x]),
                                      "ds_read_b128": len(re.findall(r"ds_read_b128", seg)), "ds_read_tr": len(re.findall(r"ds_read_b64_tr", seg)),
                                      "barriers": len(re.findall(r"s_barrier", seg)),
                                      "histogram": dict(collections.Counter(re.findall(r"^\s+([vs]_[a-z0-9_]+|ds_[a-z0-9_]+|buffer_[a-z0-9_]+)", seg, re.M)).most_common())})
    return kernels


if __name__ == "__main__":
    for f in sys.argv[1:] or ["fa_fwd_pp.hip", "fa_bwd.hip"]:
        for name, k in analyse(f).items():
            if "loops" not in k:
                continue
            worst = max(k["loops"], key=lambda d: d["mfma"]) if k["loops"] else None
            inloop = (sum(d["scratch_ops"] for d in k["loops"]), sum(d["accvgpr_moves"] for d in k["loops"]))
            short = re.sub(r"^_ZN2fa\d+", "", name)[:44]
            print(f"{f:14s} {short:44s} VGPR {k.get('vgprs'):3d} AGPR {k.get('agprs'):3d} scratch {k.get('scratch_bytes'):4d} B/lane occ {k.get('occupancy')} "
                  f"| MFMA loops {len(k['loops'])} in-loop scratch ops {inloop[0]:3d} accvgpr moves {inloop[1]:3d}" + (f" | main loop: {worst['mfma']} MFMA, {worst['valu']} VALU" if worst else ""))
