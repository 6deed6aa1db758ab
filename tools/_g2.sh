mkdir -p gpurun_out/r3c
FA_BENCH_FORCE_PG=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-extra --no-cpu-baseline > gpurun_out/r3c/torchrun_n1.json 2> gpurun_out/r3c/torchrun_n1.err; echo "n1 rc=$?"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3c/torchrun_n2_refused.json 2> gpurun_out/r3c/torchrun_n2_refused.err; echo "n2 (must refuse: 1 device) rc=$?"
grep -h "one rank per GPU" gpurun_out/r3c/torchrun_n2_refused.err | head -2
FA_BENCH_ALLOW_OVERSUBSCRIBE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3c/torchrun_n2_oversub.json 2> gpurun_out/r3c/torchrun_n2_oversub.err; echo "n2 oversubscribed rc=$?"
python - <<'PY'
import json
for f in ("torchrun_n1","torchrun_n2_oversub"):
    try:
        d=[json.loads(l) for l in open(f"gpurun_out/r3c/{f}.json") if l.startswith("{")][0]
        print(f, d["n_gpus"], round(d["value"]), d["comm_backend"], list(d["extra"].keys())[:3])
    except Exception as e: print(f, "ERR", e)
PY
