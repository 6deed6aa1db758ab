"""CPU: the one-process-per-GPU harness of bench.py (env-driven init, barrier-bracketed timing,
MAX-over-ranks time, SUM of units) under gloo with world_size 2, and the batch x head shard plan
each rank would take.  No kernels run (`--fake-step`); the GPU path uses the same `Dist` /
`timed_region` code with backend nccl (= RCCL)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world, extra=(), env_extra=None):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **(env_extra or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "5",
                                       "--warmup", "1", "--fake-step", *extra], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    return outs


def test_two_rank_gloo_timing_and_unit_aggregation():
    outs = _launch(2)
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 must print exactly one JSON line"
    assert not any(l.startswith("{") for l in outs[1][0].splitlines()), "only rank 0 prints"
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 5 and r["warmup"] == 1
    # fake step sleeps 2 ms * (1 + rank): the reported time must be the MAX over ranks (rank 1: >= 4 ms/step)
    assert r["ms_per_step"] >= 4.0
    assert r["local_ms"] <= r["ms_per_step"] * 5 + 1e-6
    # weak scaling: every rank owns b=4 x h=32 independent (batch, head) problems
    assert r["units_total"] == 2 * 4 * 32
    assert abs(r["value"] - r["units_total"] * 5 / (r["ms_per_step"] * 5e-3)) / r["value"] < 1e-6


def test_two_rank_strong_scaling_sweep_aggregation():
    """the N > 1 seqlen sweep: one b=4 x h=32 problem split by plan_shards, per-point aggregate over the MAX time, efficiency against
    rank 0 running the whole problem alone in the same run (fake kernel: time proportional to the (batch, head) units it is given)"""
    r = json.loads([l for l in _launch(2)[0][0].splitlines() if l.startswith("{")][0])
    sweep = r["extra"]["sweep_strong"]
    assert set(sweep) == {"512", "1024"}
    for key, pt in sweep.items():
        assert pt["units_total"] == 4 * 32, pt                      # the two shards cover the whole problem exactly once
        assert pt["shard"].startswith("batch [0,2) x kv heads [0,32)"), pt["shard"]
        assert pt["single_gpu_tflops_same_run"] > 0 and pt["aggregate_tflops"] > 0
        # a kernel whose time is proportional to its units scales ~linearly: efficiency near 1 (sleep granularity leaves slack)
        assert 0.4 <= pt["efficiency_vs_1gpu"] <= 1.4, pt
        assert abs(pt["efficiency_vs_1gpu"] - pt["aggregate_tflops"] / (2 * pt["single_gpu_tflops_same_run"])) < 1e-9


def test_eight_rank_gloo_weak_headline_and_strong_sweep():
    """world size 8 = the driver's largest SCALE point: weak-scaled headline (b=4 per rank -> b=32 = BASELINE configs[4]'s batch), the
    strong sweep on the batch x kv-head-halves plan (4 batch entries x 2 halves of the 32 kv heads), only rank 0 prints."""
    outs = _launch(8)
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1
    for o, _ in outs[1:]:
        assert not any(l.startswith("{") for l in o.splitlines()), "only rank 0 prints"
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["units_total"] == 8 * 4 * 32
    assert r["ms_per_step"] >= 16.0                                   # MAX over ranks: rank 7 sleeps 2 ms x 8 per step
    for pt in r["extra"]["sweep_strong"].values():
        assert pt["units_total"] == 4 * 32, pt                        # the eight shards cover the b4 x h32 problem exactly once
        assert pt["shard"].startswith("batch [0,1) x kv heads [0,16)"), pt["shard"]
        # eight sleeping processes on a small CI box: the bookkeeping identity is exact, the timing itself only loosely bounded
        assert abs(pt["efficiency_vs_1gpu"] - pt["aggregate_tflops"] / (8 * pt["single_gpu_tflops_same_run"])) < 1e-9
        assert 0.15 <= pt["efficiency_vs_1gpu"] <= 1.6, pt


def test_nccl_unavailable_falls_back_to_gloo():
    """--backend nccl on a box where RCCL cannot come up (no GPU here): every rank must agree to stay on gloo, the run must finish
    and say so in the JSON instead of dying (an nccl init failure must not lose the SCALE record)"""
    outs = _launch(2, extra=("--backend", "nccl"))
    r = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][0])
    assert r["comm_backend"].startswith("gloo (nccl"), r["comm_backend"]
    assert r["n_gpus"] == 2 and r["units_total"] == 2 * 4 * 32 and r["ms_per_step"] >= 4.0


def test_subgroup_adoption_success_branch_with_a_stand_in_backend():
    """The branch a real 8-GPU run takes and no 1-GPU box can: every rank reports its "nccl" subgroup initialised -> the gloo MIN agrees
    -> the subgroup is ADOPTED and the timed region's barriers and the MAX / SUM reductions go through it.  A gloo subgroup stands in
    for RCCL (FA_BENCH_SUBGROUP_BACKEND=gloo); the bookkeeping of the run must be what the plain gloo run gives."""
    for world in (2, 4):
        outs = _launch(world, extra=("--backend", "nccl"), env_extra={"FA_BENCH_SUBGROUP_BACKEND": "gloo"})
        r = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][0])
        assert r["comm_backend"].startswith("gloo subgroup standing in for nccl"), r["comm_backend"]
        # timed_region: 2 barriers; headline MAX + units SUM; the strong sweep adds its own -> the adopted group carried them all
        assert r["subgroup_collectives"]["barrier"] >= 2 and r["subgroup_collectives"]["all_reduce"] >= 2, r["subgroup_collectives"]
        assert r["n_gpus"] == world and r["units_total"] == world * 4 * 32
        assert r["ms_per_step"] >= 2.0 * world                       # MAX over ranks went through the subgroup: the slowest rank's time
        for pt in r["extra"]["sweep_strong"].values():
            assert pt["units_total"] == 4 * 32, pt


def test_gloo_backend_is_reported():
    r = json.loads([l for l in _launch(2, extra=("--backend", "gloo"))[0][0].splitlines() if l.startswith("{")][0])
    assert r["comm_backend"] == "gloo"


def test_single_process_fake_step():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--fake-step", "--steps", "3", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == 1 and r["units_total"] == 4 * 32


def test_gpus_flag_must_match_world_size():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--fake-step", "--gpus", "2"], env=env,
                         capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "torch.distributed.run" in (out.stderr + out.stdout)


def test_gpus_flag_needs_that_many_devices():
    """--gpus N with fewer than N visible ROCm devices must stop with a clear message before anything is allocated (here: 0 devices;
    the message for "no GPU at all" comes first)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "ROCm" in (out.stderr + out.stdout)


def test_gpus_flag_with_too_few_devices_stops_before_any_allocation():
    """behavioural form of the above for N > 1: two ranks, a torch that claims ONE visible device (patched in the child processes; there
    is no GPU here, so any allocation or kernel launch would die with a different error first): both ranks must stop with the
    device-count message"""
    port = _free_port()
    prog = ("import sys, runpy, torch;"
            "torch.cuda.is_available = lambda: True; torch.cuda.device_count = lambda: 1;"
            f"sys.argv = [{os.path.join(ROOT, 'bench.py')!r}, '--gpus', '2', '--steps', '1', '--backend', 'gloo'];"
            f"runpy.run_path({os.path.join(ROOT, 'bench.py')!r}, run_name='__main__')")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", prog], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        o, e = p.communicate(timeout=120)
        assert p.returncode != 0, (o, e)
        assert "only 1 ROCm device(s) visible" in (o + e) and "one rank per GPU is required" in (o + e), (o + e)[-1500:]


def test_clockbench_parser_reads_the_current_table_and_reports_drift():
    """bench.py reports the measured MFMA ceiling from tools/clockbench; a format drift must show up in the JSON, not vanish"""
    sys.path.insert(0, ROOT)
    import bench

    sample = ("variant                                                   min   median      max  (TFLOP/s over 5 interleaved runs)\n"
              "MFMA only, 1 wave/SIMD                                   1413     1654     1655\n"
              "MFMA only, 2 waves/SIMD                                  1600     1656     1658\n"
              "16x16x32 MFMA only, 2 waves/SIMD                         1911     1979     1984\n"
              "MFMA + 4 VALU, 2 waves/SIMD                              1357     1404     1405\n")
    got = bench.parse_clockbench(sample)
    assert got["tflops"] == 1656 and got["min"] == 1600 and "2 waves/SIMD" in got["what"]
    assert got["mfma_16x16x32"]["tflops"] == 1979          # reported beside, never as the round-to-round comparable ceiling
    old_format = "MFMA only, 2 waves/SIMD, all CUs   5.1 ms  s_memtime 8e6 ticks -> 1.6 GHz ; 1650 TFLOP/s ; 32 ticks/MFMA/SIMD\n"
    assert "error" in bench.parse_clockbench(old_format)
    assert "error" in bench.parse_clockbench("")


def test_ceiling_block_decomposes_the_gap_to_peak():
    """`roofline.ceiling`: nominal peak -> power-capped pure-MFMA rate of the kernel's own MFMA shape -> the kernel's own structure without its
    dependencies (the 16x16x32 two-group ping-pong probe) -> shipped, all from live clockbench rows of the kernel's own MFMA shape; the mixed-stream probe and the
    committed ablation ratios stand beside the chain.  Pure host arithmetic, checked here on the committed round-6 table"""
    sys.path.insert(0, ROOT)
    import bench

    with open(os.path.join(ROOT, "profiles", "r6_clockbench_instruction_mix.log")) as f:
        rows = bench.parse_clockbench_rows(f.read().split("grid 1)")[0])                 # (the chip-wide table; the one-CU table follows it in the file)
    assert rows[bench.PROBE_PURE16] > rows["MFMA only, 2 waves/SIMD"] > 1000
    for k in (bench.PROBE_MIX_D128, bench.PROBE_PP_D128, bench.PROBE_MIX_D64, bench.PROBE_PP_D64):
        assert 800 < rows[k] < rows[bench.PROBE_PURE16], k                                 # the labels bench.py looks up exist in tools/clockbench's table
    c = bench.ceiling_block("c3", "fa_fwd_pp16_kernel", 1260.0, rows)
    assert c["pure_mfma_on_n01_operands_tflops"]["this_kernels_mfma_shape"] == rows[bench.PROBE_PURE16] and c["frac_of_power_capped_mfma_rate"] == 1260.0 / rows[bench.PROBE_PURE16]
    assert c["chain_tflops"][0][1] == 2500.0 and c["chain_tflops"][-1][1] == 1260.0 and len(c["chain_tflops"]) == 4
    assert c["chain_tflops"][2][1] == rows[bench.PROBE_PP_D128] and c["shipped_over_structure_probe"] == 1260.0 / rows[bench.PROBE_PP_D128]
    assert c["shipped_over_mixed_stream_probe"] == 1260.0 / rows[bench.PROBE_MIX_D128]
    vals = [v for _, v in c["chain_tflops"]]
    assert vals == sorted(vals, reverse=True), vals                                   # every step of the chain costs something
    prod = 1.0
    for r in c["chain_step_ratios"]:
        prod *= r
    assert abs(prod - 1260.0 / 2500.0) < 1e-12
    if "same_structure_ablated" in c:                                                   # (a committed profiles/rNN_fwd_ceiling_ablations.json exists)
        assert 0.5 < c["same_structure_ablated"]["time_ratio_vs_shipped"]["no_lds_reads_no_exp_no_dma"] < 1.0
    c32 = bench.ceiling_block("c3", "fa_fwd_pp_kernel", 1200.0, rows)                   # the 32x32x16 kernel: its own pure-MFMA row, no 16x16x32 structure probe
    assert c32["pure_mfma_on_n01_operands_tflops"]["this_kernels_mfma_shape"] == rows["MFMA only, 2 waves/SIMD"] and len(c32["chain_tflops"]) == 3
    assert bench.ceiling_block("c3", "fa_fwd_pp16_kernel", 1260.0, None)["chain_tflops"][0][1] == 2500.0      # clockbench missing: still a block


def _driver_scale_command(n, extra=()):
    """the driver's SCALE launch, verbatim (task contract): python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1", "--fake-step", *extra]


def test_driver_scale_command_line_at_eight_ranks():
    """the first real SCALE run should debut RCCL and nothing else.  bench.py is launched exactly as the driver launches it
    (torch.distributed.run spawns the ranks and sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; no environment prepared by this test), with the kernels
    replaced by --fake-step: ONE JSON line on the job's stdout, n_gpus = 8, the headline workload = BASELINE configs[2] per GPU (per-GPU work fixed:
    `scaling: weak`, global batch 32), configs[4] (b = 32 = 4 per rank, 16k, non-causal) named in `extra`, and at N = 1 the same launcher gives the
    line a plain `python bench.py` gives (the N = 1 value path does not depend on how it was launched)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run(_driver_scale_command(8), env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["steps"] == 3 and r["warmup"] == 1
    assert r["scaling"] == "weak" and r["higher_is_better"] is True and r["vs_baseline"] is None
    assert r["config"]["workload"].startswith("BASELINE configs[2] per GPU: fwd b=4 seq=16384 h=32 h_k=32 d=128 fp16 causal")
    assert r["config"]["global_batch"] == 32 and r["config"]["parallelism"] == "batch-sharded x8, no collective"
    assert r["extra"]["c5_weak_noncausal_16k"]["config"] == "BASELINE configs[4] shape: fwd b=32 (4 per rank) seq=16384 h=32 d=128 fp16 non-causal"
    assert r["units_total"] == 8 * 4 * 32 and r["ms_per_step"] >= 16.0           # MAX over ranks (rank 7 sleeps 2 ms x 8 per fake step)
    assert r["comm_backend"].startswith("gloo")                                  # (no GPU here: the nccl adoption reports why not)
    # N = 1: launcher or not, the same line (timing fields aside)
    one = subprocess.run(_driver_scale_command(1), env=env, capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stderr[-3000:]
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--fake-step"], env=env,
                           capture_output=True, text=True, timeout=120)
    assert plain.returncode == 0, plain.stderr[-3000:]
    a, b = (json.loads([l for l in o.stdout.splitlines() if l.startswith("{")][0]) for o in (one, plain))
    timing = {"value", "ms_per_step", "local_ms", "extra", "comm_backend", "subgroup_collectives"}
    assert {k: v for k, v in a.items() if k not in timing} == {k: v for k, v in b.items() if k not in timing}
    assert a["n_gpus"] == b["n_gpus"] == 1 and a["config"]["global_batch"] == 4 and "c5_weak_noncausal_16k" not in a["extra"]
