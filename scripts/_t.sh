cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_heat.py -x -q -k "deferred or heat" > gpurun_out/t1.log 2>&1; tail -12 gpurun_out/t1.log | cut -c1-600
if grep -q failed gpurun_out/t1.log; then exit 1; fi
for v in 0 16; do
NLG_PCG_DEFER_X=$v python3 bench.py --nel 40,25,20 --lx1 10 --ifheat --no-history --kdim 128 --steps 3 --warmup 2 --no-cpu --no-units 2>>gpurun_out/cfg4.err | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 single defer=$v', round(j['value'],3), 'matvecs/s', round(j['ms_per_step'],2))"
done
python3 bench.py --nel 40,25,20 --lx1 10 --ifheat --no-history --kdim 128 --block 4 --steps 2 --warmup 1 --no-cpu --no-units 2>>gpurun_out/cfg4.err > gpurun_out/r04_cfg4_block4.json; python3 -c "import json; j=json.loads(open('gpurun_out/r04_cfg4_block4.json').read().strip().splitlines()[-1]); print('cfg4 block4', round(j['value'],3), 'matvecs/s', round(j['ms_per_step'],2), 'ms per block step', j['config']['launches_per_vector'], 'launches/vector')"
