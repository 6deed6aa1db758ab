mkdir -p gpurun_out/r3c
FA_BENCH_ALLOW_OVERSUBSCRIBE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3c/torchrun_n2_oversub.json 2> gpurun_out/r3c/torchrun_n2_oversub.err; echo "n2 oversubscribed rc=$?"
FA_BENCH_FORCE_PG=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-extra --no-cpu-baseline > gpurun_out/r3c/torchrun_n1.json 2> gpurun_out/r3c/torchrun_n1.err; echo "n1 rc=$?"
python - <<'PY'
import json
for f in ("torchrun_n1","torchrun_n2_oversub"):
    d=[json.loads(l) for l in open(f"gpurun_out/r3c/{f}.json") if l.startswith("{")][0]
    print(f, d["n_gpus"], round(d["value"]), d["comm_backend"][:40], d["roofline"]["power_and_sclk"]["timed_region"].get("power_w_mean"), d.get("git_commit"))
PY
