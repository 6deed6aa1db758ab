"""Guard against silent register-allocation pathologies in the HIP kernels (runs on CPU: hipcc cross-compiles gfx950).

Parity tests cannot see a kernel that spills or shuffles its accumulators inside the hot loop -- it is merely slow.  Round 1
shipped a causal dK/dV kernel with 258 v_accvgpr moves + 48 scratch ops per loop iteration for most of the round (1.65x slower
backward on causal shapes, profiles/r1_bwd_causal_ab.log) because only the non-causal instance had been inspected.  This test
compiles every kernel instantiation and fails on scratch traffic or accumulator shuffles inside any MFMA loop."""
import pytest

from _kernel_isa import analyse

FILES = ["fa_fwd_pp.hip", "fa_fwd_pp16.hip", "fa_bwd.hip", "fa_bwd_dq16.hip", "fa_bwd_dkdv16.hip"]
# whole-kernel scratch that is known, outside every loop (prologue / epilogue), and bounded here so growth is noticed
SCRATCH_ALLOWED = {"fa_bwd_dkdv_kernel": 112, "fa_bwd_dq_kernel": 64, "fa_bwd_dkdv16_kernel": 32}      # dK/dV: D = 64 causal carries 100 B since the second (workspace) epilogue
# scratch ops INSIDE an MFMA loop: zero everywhere since round 3 (rounds 1-2 allowed the two D = 64 backward kernels, squeezed to 128
# registers for two workgroups per CU, 1-4 spill ops per tile).  (kernel substring, head-dim substring) -> max ops per loop.
# Accumulator shuffles: never.
INLOOP_SCRATCH_ALLOWED = {}


@pytest.fixture(scope="module")
def kernels():
    out = {}
    for f in FILES:
        for name, info in analyse(f).items():
            out[(f, name)] = info
    return out


def test_every_kernel_was_analysed(kernels):
    with_loops = [k for k, v in kernels.items() if v.get("loops")]
    assert len(kernels) >= 20 and len(with_loops) >= 16, (len(kernels), len(with_loops))
    for (f, name), info in kernels.items():
        assert {"vgprs", "agprs", "scratch_bytes", "occupancy"} <= set(info), (f, name, info)
        # no vacuous passes: every MFMA kernel must have been seen WITH its MFMAs and at least one MFMA loop
        if "dot_do_o" not in name and "sum_splits" not in name:      # the two HBM-bound helpers hold no MFMA
            assert info.get("mfma_total", 0) >= 16 and info.get("loops"), (f, name, info.get("mfma_total"), info.get("loops"))


def test_no_spills_or_accumulator_shuffles_inside_mfma_loops(kernels):
    bad = []
    for (f, name), info in kernels.items():
        for loop in info.get("loops", []):
            limit = next((v for (k, dd), v in INLOOP_SCRATCH_ALLOWED.items() if k in name and dd in name), 0)
            if loop["scratch_ops"] > limit or loop["accvgpr_moves"]:
                bad.append((f, name, loop, limit))
    assert not bad, bad


def test_whole_kernel_scratch_is_zero_or_on_the_allow_list(kernels):
    for (f, name), info in kernels.items():
        limit = next((v for k, v in SCRATCH_ALLOWED.items() if k in name), 0)
        assert info["scratch_bytes"] <= limit, (f, name, info["scratch_bytes"], limit)


def test_two_waves_per_simd_for_the_eight_wave_kernels(kernels):
    for (f, name), info in kernels.items():
        if any(k in name for k in ("fa_fwd_pp_kernel", "fa_fwd_pp16_kernel", "fa_bwd_dkdv_kernel", "fa_bwd_dq_kernel", "fa_bwd_dq16_kernel", "fa_bwd_dkdv16_kernel")):
            assert info["occupancy"] >= 2, (f, name, info["occupancy"])


def test_d64_kernels_fit_two_workgroups_per_cu(kernels):
    """D = 64 instances must reach 4 waves per SIMD = two 8-wave workgroups per CU (their LDS allows it); losing it cost 7-29 % forward
    and 17-22 % backward time in round 1 (profiles/r1_fwd_d64_occupancy_ab.log, r1_bwd_d64_occupancy_ab.log) with parity unaffected"""
    seen = 0
    for (f, name), info in kernels.items():
        # (the D = 64 FORWARD has a second, 128-key shape with one workgroup per CU since round 3: profiles/r3_fwd_d64_tile_ab.log)
        narrow = "fa_fwd_pp_kernel" in name and "Li64ELb" in name and "Li128E" not in name          # <T, 64, CAUSAL, BN = 64>
        if ("Li64E" in name and any(k in name for k in ("fa_bwd_dq_kernel", "fa_bwd_dkdv_kernel"))) or narrow:
            seen += 1
            assert info["occupancy"] >= 4, (f, name, info["occupancy"], info["vgprs"], info["agprs"])
            assert 2 * info["lds_bytes"] <= 160 * 1024, (f, name, info["lds_bytes"])
    assert seen == 12, seen
    wide = [n for (f, n) in kernels if "fa_fwd_pp_kernel" in n and "Li64ELb" in n and "Li128E" in n]
    assert len(wide) == 4, wide                                                                      # <T, 64, CAUSAL, BN = 128>


def test_kernels_touch_m0_only_in_their_own_lds_dma_statements(kernels):
    """every kernel issues its LDS-DMA from inline asm WITHOUT saving / restoring M0 (dma16_to_lds_hidden<false>; the backward ones since
    round 4, FA_BWD_DMA_SAVE_M0 = 0); that is only sound while nothing hipcc generates for those kernels reads or writes M0 (no
    compiler-visible LDS-DMA, no indirect register indexing)"""
    fwd = bwd = 0
    for (f, name), info in kernels.items():
        if "fa_fwd_pp_kernel" in name or "fa_fwd_pp16_kernel" in name:
            fwd += 1
            assert info["m0_outside_asm"] == 0, (name, info["m0_outside_asm"])
        if any(k in name for k in ("fa_bwd_dq_kernel", "fa_bwd_dq16_kernel", "fa_bwd_dkdv_kernel", "fa_bwd_dkdv16_kernel")):
            bwd += 1
            assert info["m0_outside_asm"] == 0, (name, info["m0_outside_asm"])
    assert fwd >= 16 and bwd >= 24, (fwd, bwd)


def test_mfma16_forward_keeps_its_accumulators_in_place(kernels):
    """fa_fwd_pp16: hipcc does not tie a 4-pass MFMA's destination to its C operand, and a conditional rescale of the 64 O registers that
    merges back into the hot path made it copy half of them per tile (16 v_mov_b64 + 19 v_mov_b32, +44 % VALU instructions, 5 % slower
    than the 32x32x16 kernel instead of 5 % faster: profiles/r3_fwd_mfma16_ab.log); with the SLP vectoriser on, it spills as well.  The
    steady-state loop must hold no 64-bit register copy and only the handful of v_mov_b32 the cross-lane max needs."""
    seen = 0
    for (f, name), info in kernels.items():
        if "fa_fwd_pp16_kernel" in name:
            seen += 1
            main = max(info["loops"], key=lambda l: l["mfma"])
            # three tiles of 64; the fp16 instances sum their softmax rows in the matrix pipe (round 4: +4 ones-row MFMAs per tile, and the
            # row-sum adds must be gone from the loop - only the ~6 address / bookkeeping adds remain)
            # head_dim 64 (128-key tiles, round 4): the same 64 MFMAs per tile, +8 ones-row MFMAs (four 32-key chunks)
            rowsum, d64 = "IDF16_" in name, "Li64ELb" in name
            assert main["mfma"] == ((216 if d64 else 204) if rowsum else 192), (name, main["mfma"])
            if rowsum:
                assert main["histogram"].get("v_add_f32_e32", 0) <= 12 and main["histogram"].get("v_pk_maximum3_f16", 0) == (48 if d64 else 24), (name, main["histogram"])
            assert main["histogram"].get("v_mov_b64_e32", 0) == 0, (name, main["histogram"])
            assert main["histogram"].get("v_mov_b32_e32", 0) <= 24, (name, main["histogram"])
    assert seen == 8


def test_dkdv16_ring_addresses_are_toggled_not_recomputed(kernels):
    """fa_bwd_dkdv16: the Q / dO ring slot is a run-time value, and base + slot offset per fragment read was a third of the tile loop's
    VALU instructions (34 of 101); the base registers are absolute addresses flipped to the other slot by one inline-asm v_xor each per tile
    (-2..-6 % of the kernel, profiles/r3_bwd_mfma16_ab.log).  hipcc puts the adds back if it can see through the toggle: keep it honest."""
    seen = 0
    for (f, name), info in kernels.items():
        if "fa_bwd_dkdv16_kernel" in name:
            seen += 1
            main = max(info["loops"], key=lambda l: l["mfma"])
            d64 = "Li64ELb" in name                                                                     # round 5: the same kernel at head_dim 64
            assert main["mfma"] == (32 if d64 else 64), (name, main["mfma"])
            assert main["histogram"].get("v_xor_b32", 0) == (7 if d64 else 13), (name, main["histogram"])      # KS row + DB transposed + 1 statistics base
            assert main["histogram"].get("v_add_u32_e32", 0) <= 14, (name, main["histogram"])
    assert seen == 8


def test_d64_mfma16_backward_kernels_are_clean(kernels):
    """round 5: fa_bwd_dq16 / fa_bwd_dkdv16 at head_dim 64 run ONE workgroup per compute unit (160-180 registers; at the 128-register budget of two
    co-resident workgroups dQ spills 46-86 registers) - they must at least be free of scratch and fit that one workgroup"""
    seen = 0
    for (f, name), info in kernels.items():
        if "Li64ELb" in name and any(k in name for k in ("fa_bwd_dq16_kernel", "fa_bwd_dkdv16_kernel")):
            seen += 1
            assert info["occupancy"] >= 2 and info["lds_bytes"] <= 160 * 1024, (name, info["occupancy"], info["lds_bytes"])
            assert info["scratch_bytes"] == 0, (name, info["scratch_bytes"])
    assert seen == 8, seen


def test_no_instruction_touches_an_mfma_result_before_it_has_landed(kernels):
    """Round 5: an experiment build of the 16x16x32 forward (three query columns per wave) computed wrong values because hipcc scheduled a v_fma_f32 that
    reads a score two instructions behind the inline-asm MFMA producing it - above the barrier and the s_nop pad, whose "memory" clobber does not order
    register-only instructions (profiles/r5_fwd_qb3_ab.log).  Every pad now names its registers; this walks the ISA of every shipped instance."""
    bad = {k: v["mfma_hazards"][:3] for k, v in kernels.items() if v.get("mfma_hazards")}
    assert not bad, bad


def test_three_query_column_experiment_build_stays_buildable_and_hazard_free():
    """The switches of the round-5 experiment stay in the source (FA_FWD_D128_QB / FA_FWD_D128_BN, profiles/r5_fwd_qb3_ab.log): the build they select must keep
    compiling at two waves per SIMD inside 160 KiB of LDS, without scratch in its MFMA loops, and - it is the build the hazard was found in - without a finding."""
    ks = {n: i for n, i in analyse("fa_fwd_pp16.hip", extra_flags=["-DFA_FWD_D128_QB=3", "-DFA_FWD_D128_BN=32"]).items() if "Li128ELb" in n and "Li32ELi3E" in n}
    assert len(ks) == 4, list(ks)
    for name, info in ks.items():
        assert info["occupancy"] >= 2 and info["lds_bytes"] <= 160 * 1024, (name, info["occupancy"], info["lds_bytes"])
        assert info["scratch_bytes"] <= 32, (name, info["scratch_bytes"])          # (fp16 causal: four registers stored before the loops, reloaded behind them)
        assert all(l["scratch_ops"] == 0 for l in info["loops"]), name
        assert info["mfma_hazards"] == [], (name, info["mfma_hazards"][:3])


def test_hazard_scan_sees_the_round5_pathology_and_hipccs_own_spacing():
    from _mfma_hazards import scan_kernel

    racy = """k:
	v_mfma_f32_16x16x32_f16 v[170:173], v[196:199], v[58:61], v[170:173]
	v_mfma_f32_16x16x32_f16 v[162:165], v[196:199], v[62:65], v[162:165]
	s_nop 0
	v_fma_f32 v208, s40, v170, v187
	s_barrier
	s_nop 7
	v_fma_f32 v209, s40, v162, v187""".split("\n")
    v = scan_kernel(racy, 0, len(racy))
    assert len(v) == 1 and "v170" in v[0][1]
    # what hipcc itself leaves behind a builtin MFMA of either shape (fa_bwd_dq16 / fa_bwd): accepted
    ok = """k:
	v_mfma_f32_16x16x32_f16 v[116:119], v[126:129], v[2:5], v[46:49]
	v_exp_f32_e32 v98, v42
	v_fma_f32 v42, v44, s2, -v104
	v_exp_f32_e32 v107, v42
	v_fma_f32 v42, v45, s2, -v104
	v_exp_f32_e32 v108, v42
	s_nop 2
	v_fma_f32 v42, v116, s2, -v103
	v_mfma_f32_32x32x16_f16 v[34:49], v[54:57], v[82:85], v[34:49]
	s_nop 11
	v_fma_f32 v55, v34, s0, -v118""".split("\n")
    assert scan_kernel(ok, 0, len(ok)) == []
    # an AGPR accumulator read one state early
    early = """k:
	v_mfma_f32_32x32x16_f16 a[0:15], v[54:57], v[82:85], a[0:15]
	s_nop 10
	v_accvgpr_read_b32 v1, a3""".split("\n")
    assert len(scan_kernel(early, 0, len(early))) == 1
    # (round 6) a reader at a LOOP TOP behind an MFMA at the loop's tail: seen through the back edge, and only through it
    loop = """k:
.LBB0_1:
	v_add_f32 v9, v0, v0
	s_nop 7
	s_nop 7
	v_mfma_f32_16x16x32_f16 v[0:3], v[4:7], v[4:7], v[0:3]
	s_cbranch_vccnz .LBB0_1
	s_nop 7
	v_add_f32 v9, v0, v0""".split("\n")
    v = scan_kernel(loop, 0, len(loop))
    assert len(v) == 1 and v[0][0] == 3 and v[0][4] == 1          # line 3 (the loop's first instruction), one wait state behind the MFMA (the branch)
    spaced = [l for l in loop]
    spaced.insert(2, "\ts_nop 7")                                  # eight states at the loop top: clean
    assert scan_kernel(spaced, 0, len(spaced)) == []


def test_guard_detects_the_known_pathology():
    """the detector must fire on the construct it exists for: the wave-level skip branch around asm-accumulator MFMAs"""
    ks = analyse("fa_bwd.hip", extra_flags=["-DFA_TEST_DKDV_SKIP_BRANCH"])
    causal_d128 = [v for n, v in ks.items() if "fa_bwd_dkdv_kernel" in n and "Li128ELb1" in n]
    assert causal_d128
    for info in causal_d128:
        assert sum(l["accvgpr_moves"] for l in info["loops"]) > 100, info["loops"]
