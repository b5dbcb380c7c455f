cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_edge_cases.py -x -q -k "compact or deferred" > gpurun_out/t1.log 2>&1; tail -12 gpurun_out/t1.log | cut -c1-600
if grep -q failed gpurun_out/t1.log; then exit 1; fi
python3 -m pytest tests/test_gpu_n8.py tests/test_gpu_n10.py tests/test_gpu_linop.py tests/test_gpu_block.py tests/test_gpu_proj.py -x -q > gpurun_out/t2.log 2>&1; tail -5 gpurun_out/t2.log | cut -c1-300
for v in 0 16 0 16; do
  NLG_PCG_DEFER_XP=$v python3 bench.py --steps 8 --warmup 2 --no-cpu --no-units 2>>gpurun_out/hack.err | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('E10k deferxp=$v', round(j['ms_per_step'],3), 'pprec ms/step', r['class_ms_per_step'].get('pprec'), 'vec_ops', r['class_ms_per_step'].get('vec_ops'), 'p its', j['config']['pressure_iters_per_time_step'])"
done
