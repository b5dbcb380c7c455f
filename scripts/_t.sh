cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_n8.py tests/test_gpu_ops.py -x -q > gpurun_out/t2.log 2>&1; tail -3 gpurun_out/t2.log | cut -c1-300
for v in 1 2; do
python3 bench.py --steps 8 --warmup 2 --no-cpu --no-units 2>>gpurun_out/hack.err | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('E10k', round(j['ms_per_step'],3), 'conv ms/step', r['class_ms_per_step'].get('conv'), r['class_launches_per_step'].get('conv'))"
NLG_CONV_MFMA=0 python3 bench.py --steps 4 --warmup 2 --no-cpu --no-units 2>>gpurun_out/hack.err | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('E10k vector-pipe conv', round(j['ms_per_step'],3), 'conv ms/step', r['class_ms_per_step'].get('conv'))"
done
