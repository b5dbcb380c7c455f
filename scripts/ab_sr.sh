# single-reduction PCG: parity (two processes), the multi-rank tests (on by default there), cost on one rank, counts of the 2-rank rehearsal
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_multirank.py tests/test_gpu_bench_multi.py tests/test_gpu_heat.py -x -q 2>&1 | tail -4 &&
for v in 0 1; do
  NLG_PCG_SINGLE_RED=$v python3 bench.py --steps 6 --warmup 2 --no-cpu --no-units 2>>gpurun_out/ab_sr.err | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print('E10k SR=$v', round(j['ms_per_step'],3), 'launches', c['launches_per_step'], 'collectives', c['collectives_per_step'], 'v its', c['helmholtz_iters_per_time_step'])"
  NLG_PCG_SINGLE_RED=$v python3 bench.py --nel 13,10,10 --steps 10 --warmup 3 --no-cpu --no-units 2>>gpurun_out/ab_sr.err | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print('E1300 SR=$v', round(j['ms_per_step'],3), 'launches', c['launches_per_step'], 'collectives', c['collectives_per_step'])"
done
NLG_PCG_SINGLE_RED=1 timeout -k 10 500 python3 bench.py --gpus 2 --transport shm --steps 2 --warmup 1 --no-cpu --no-units > gpurun_out/r04_bench_2ranks_shm_one_gpu.json 2>gpurun_out/r04_bench_2ranks.err
python3 -c "import json; j=json.loads(open('gpurun_out/r04_bench_2ranks_shm_one_gpu.json').read().strip().splitlines()[-1]); c=j['config']; print('2 ranks shm: launches/step', c['launches_per_step'], 'collectives/step', c['collectives_per_step'], 'p its', c['pressure_iters_per_time_step'], 'v its', c['helmholtz_iters_per_time_step'])"
