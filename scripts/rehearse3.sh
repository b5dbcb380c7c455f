#!/bin/bash
# 3 ranks x 16^3 elements on one GPU (shm transport), aggregated coarse level; prints iterations per time step
# usage: rehearse3.sh <NLG_IFACE_AGG mode> [NLG_LIB_INVERSE_MIN]
mode=$1
export NLG_IFACE_AGG=$mode NLG_LIB_INVERSE_MIN=${2:-4096}
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 3 --steps 2 --warmup 1 --transport shm --nel 16,16,16 --no-units > gpurun_out/rehearse3_$mode.log 2>&1
echo "mode $mode libmin $NLG_LIB_INVERSE_MIN: $(grep -o 'pressure_iters_per_time_step": [0-9.]*\|setup_s": [0-9.]*' gpurun_out/rehearse3_$mode.log | tr '\n' ' ')"
