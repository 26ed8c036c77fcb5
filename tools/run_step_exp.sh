set -u
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export RT_BENCH_FORCE_MULTI=1 MASTER_ADDR=127.0.0.1
( time timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29581 bench.py --gpus 1 --steps 3 --warmup 1 > gpurun_out/bench_multi1.json 2> gpurun_out/bench_multi1.err ) 2>&1 | tail -4
echo "rc=$?"; tail -n 12 gpurun_out/bench_multi1.err; python3 - <<'PY'
import json
for l in open("gpurun_out/bench_multi1.json"):
    if l.startswith("{"):
        d=json.loads(l)
        print({k:d[k] for k in ("value","n_gpus","ms_per_step")}, d["config"].get("assembled_frame_equals_one_gpu_frame"), d.get("failed_lines"))
        for k,v in d.get("secondary",{}).items():
            print(k, {a:b for a,b in v.items() if not isinstance(b,(dict,list))} if isinstance(v,dict) else v, (v.get("config") if isinstance(v,dict) else None))
PY
