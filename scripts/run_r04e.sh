cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_n8.py tests/test_gpu_multirank.py -x -q 2>&1 | tail -5 &&
bash scripts/ab_small.sh > gpurun_out/r04_final_ab_small.txt 2>&1; cat gpurun_out/r04_final_ab_small.txt
python3 bench.py --nel 13,10,10 --steps 10 --warmup 3 --no-cpu --no-units 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('E1300 p its', j['config']['pressure_iters_per_time_step'], 'v its', j['config']['helmholtz_iters_per_time_step'])"
NLG_SMALL_E=0 python3 bench.py --nel 13,10,10 --steps 10 --warmup 3 --no-cpu --no-units 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('E1300 SMALL_E=0 p its', j['config']['pressure_iters_per_time_step'], round(j['ms_per_step'],3))"
for n in 4; do
  timeout -k 10 500 python3 bench.py --gpus $n --transport shm --steps 2 --warmup 1 --no-cpu --no-units > gpurun_out/r04_bench_${n}ranks_shm_one_gpu.json 2>gpurun_out/r04_bench_${n}ranks.err
  python3 -c "import json; j=json.loads(open('gpurun_out/r04_bench_${n}ranks_shm_one_gpu.json').read().strip().splitlines()[-1]); c=j['config']; print('$n ranks shm: launches/step', c['launches_per_step'], 'collectives/step', c['collectives_per_step'], 'p its', c['pressure_iters_per_time_step'], 'v its', c['helmholtz_iters_per_time_step'])"
done
# k = 128 orthogonalisation: two-tile fused sweep against the round-3 path (fused over the last 64 only)
for f in 128 64; do
NLG_CGS2_FUSE_MAX=$f python3 bench.py --nel 20,20,10 --lx1 10 --ifheat --no-history --kdim 128 --steps 3 --warmup 2 --no-cpu 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FUSE_MAX=$f E4000 lx10 m128: ortho ms at k=m', j['config']['arnoldi_orthogonalisation_ms_at_k=m'], 'ms/step', round(j['ms_per_step'],2))"
done
