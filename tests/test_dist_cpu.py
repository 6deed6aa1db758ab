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
