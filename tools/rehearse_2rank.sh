#!/bin/bash
# The N > 1 code path of bench.py (headline + the secondary workloads of N > 1) rehearsed with two ranks sharing one GPU over gloo.
cd $GRAFT_REPO_ROOT
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --single-device --steps 5 --warmup 2 > gpurun_out/r3_2rank_secondary_v4.json 2> gpurun_out/r3_2rank_secondary_v4.err; echo rc=$?
grep "^{" gpurun_out/r3_2rank_secondary_v4.json | python3 -c '
import json,sys
d=json.loads(sys.stdin.read())
print(d["value"], d["n_gpus"], d["ms_per_step"], list(d.get("secondary",{}).keys()), [round(s["value"],1) for s in d.get("secondary",{}).values()], "cpu_baseline" in d)'
