#!/bin/bash
# Round 4, lease 19: the torchrun launch path (one-rank RCCL group: on-device weight broadcast + gather) on the final tree
O=gpurun_out/r04t; mkdir -p $O
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29641 bench.py --gpus 1 --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-batched > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; echo "torchrun rc=$? $(date +%T)"
python - "$O/bench_torchrun1.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
    print('value', round(d['value'],4), 'n_gpus', d['n_gpus'], 'process_group', c.get('process_group'), 'bcast_s', c.get('weights_broadcast_s'), 'gathered', c.get('gathered_latents'))
except Exception as e: print('unreadable', e)
PY
tail -4 $O/bench_torchrun1.err
