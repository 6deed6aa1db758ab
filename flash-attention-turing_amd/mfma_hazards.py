"""Static scan for the one hazard class hipcc cannot see in these kernels: the accumulating MFMAs are inline asm ("+v" / "+a" tiles, fa_device.hpp), so the
compiler's hazard recogniser does not know their results land passes later, and it is free to schedule a register-only VALU instruction that reads such a
result directly behind the MFMA - above the `asm volatile("s_nop ...")` pad, whose "memory" clobber orders memory operations only.  Round 5 met exactly that
in an experiment build (three query columns per wave, profiles/r5_fwd_qb3_ab.log: v_fma_f32 two instructions behind the MFMA that produced its input, values
wrong by ~0.1); since then every pad names the registers it protects, and this scan walks the ISA of every instance.

The scan is straight-line and conservative: every instruction counts one wait state (s_nop N: N + 1), labels are fall-through, branches keep the
pending set.  A result register touched (read or written) by a non-MFMA instruction, or read as A / B by another MFMA, fewer than NEED wait states behind
the MFMA that writes it, is reported.  NEED is calibrated on what hipcc itself inserts behind its own (builtin) MFMAs in these files: a dependent VALU sits
8 wait states behind a v_mfma_f32_16x16x32 (5 instructions + s_nop 2 in fa_bwd_dq16) and 12 behind a v_mfma_f32_32x32x16 (s_nop 11 in fa_bwd)."""
import re

NEED = {"16x16x32": 8, "32x32x16": 12}      # other shapes: 20 (none is issued from inline asm here)


def _regs(tok):
    out = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b", tok):
        if m.group(1):
            out.update((m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def scan_kernel(lines, start, end):
    pending, viol = [], []     # pending: (dst regs, wait states elapsed, needed, line no, text)
    for i in range(start + 1, end):
        l = lines[i].split(";")[0].strip()
        if not l or l.endswith(":") or (l.startswith(".") and not l.startswith(".LBB")):
            continue
        op = l.split()[0]
        ws = 1
        if op == "s_nop":
            ws = int(l.split()[1]) + 1
        elif op.startswith("v_mfma") or op.startswith("v_smfmac"):
            need_here = next((n for s, n in NEED.items() if s in op), 20)
            ops = l[len(op):].split(",")
            dst, srcab = _regs(ops[0]), _regs(ops[1]) | _regs(ops[2])
            for d, e, need, ln, t in pending:
                if d & srcab and e < need:
                    viol.append((i + 1, l, ln + 1, t, e, need))
            pending = [(d, e + 1, need, ln, t) for d, e, need, ln, t in pending if e + 1 < 48]
            pending.append((dst, 0, need_here, i, l))
            continue
        elif op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            pending = [(d, e + 1, need, ln, t) for d, e, need, ln, t in pending]
            continue
        if not op.startswith("s_"):
            r = _regs(l[len(op):])
            for d, e, need, ln, t in pending:
                if d & r and e < need:
                    viol.append((i + 1, l, ln + 1, t, e, need))
        pending = [(d, e + ws, need, ln, t) for d, e, need, ln, t in pending if e + ws < 48]
    return viol


def scan_file(path):
    """{kernel name: [violations]} for every kernel of a --cuda-device-only -S listing"""
    lines = open(path).read().split("\n")
    out = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", l)
        if not m:
            continue
        end = next((j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end")), len(lines))
        out[m.group(1)] = scan_kernel(lines, i, end)
    return out


if __name__ == "__main__":
    import sys

    total = 0
    for k, v in scan_file(sys.argv[1]).items():
        print(k, len(v))
        for x in v[:12]:
            print("   line %d: %s   <- line %d: %s   (%d of %d wait states)" % x)
        total += len(v)
    print("violations:", total)
