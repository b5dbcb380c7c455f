#!/bin/bash
# debug helper: 2 ranks on one GPU, aggregated coarse level forced
export NLG_COARSE_EXACT_MAX=50 NLG_PPREC_DEBUG=1
mkdir -p /tmp/mr
python tests/multirank_worker.py 0 2 /nlg_dbg1 /tmp/mr agg3d > /tmp/mr/r0.log 2>&1 &
P0=$!
python tests/multirank_worker.py 1 2 /nlg_dbg1 /tmp/mr agg3d > /tmp/mr/r1.log 2>&1
wait $P0
grep pprec /tmp/mr/r0.log /tmp/mr/r1.log
