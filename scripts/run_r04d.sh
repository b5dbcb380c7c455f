# round-4 closing numbers: proxy pair on one box, partitioned rehearsals (counts only), block of 4
cd $GRAFT_REPO_ROOT
bash scripts/ab_small.sh > gpurun_out/r04_final_ab_small.txt 2>&1; cat gpurun_out/r04_final_ab_small.txt
for n in 2 4; do
  timeout -k 10 500 python3 bench.py --gpus $n --transport shm --steps 2 --warmup 1 --no-cpu --no-units > gpurun_out/r04_bench_${n}ranks_shm_one_gpu.json 2>gpurun_out/r04_bench_${n}ranks.err
  python3 -c "import json; j=json.loads(open('gpurun_out/r04_bench_${n}ranks_shm_one_gpu.json').read().strip().splitlines()[-1]); c=j['config']; print('$n ranks shm: launches/step', c['launches_per_step'], 'collectives/step', c['collectives_per_step'], 'p its', c['pressure_iters_per_time_step'], 'v its', c['helmholtz_iters_per_time_step'])"
done
python3 bench.py --block 4 --steps 4 --warmup 2 --no-cpu --no-units 2>/dev/null > gpurun_out/r04_bench_blk4.json; python3 -c "import json; j=json.loads(open('gpurun_out/r04_bench_blk4.json').read().strip().splitlines()[-1]); print('block 4', round(j['value'],2), 'matvecs/s', j['config']['launches_per_vector'], 'launches/vector', j['roofline']['kernel'], round(j['roofline']['frac'],3))"
