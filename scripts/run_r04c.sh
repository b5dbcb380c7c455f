# heat lanes parity, IPC probe, Fortran + Newton tests, block / heat benches
cd $GRAFT_REPO_ROOT
python3 scripts/ipc_probe.py > gpurun_out/r04_ipc_probe.txt 2>&1; tail -3 gpurun_out/r04_ipc_probe.txt
python3 -m pytest tests/test_gpu_heat.py tests/test_gpu_fortran.py tests/test_gpu_reference_data.py tests/test_gpu_block.py -x -q 2>&1 | tail -5 &&
for b in 1 4; do
python3 bench.py --nel 20,10,10 --lx1 10 --ifheat --no-history --kdim 64 --block $b --steps 3 --warmup 2 --no-cpu --no-units 2>>gpurun_out/r04c.err | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('heat lx10 E2000 block $b', round(j['value'],3), 'matvecs/s', round(j['ms_per_step'],2), 'ms/step', j['config']['launches_per_vector'], 'launches/vector')"
done
