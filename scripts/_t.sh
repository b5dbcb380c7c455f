cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_n10.py -x -q > gpurun_out/t2.log 2>&1; tail -3 gpurun_out/t2.log | cut -c1-300
for v in 0 1 0 1; do
NLG_AXHELM_XCD=$v python3 bench.py --lx1 10 --steps 5 --warmup 2 --no-cpu --no-units 2>>gpurun_out/hack.err | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('lx1=10 xcd=$v', round(j['ms_per_step'],3), round(j['value'],3), 'axhelm', r['class_ms_per_step'].get('axhelm'), r['class_launches_per_step'].get('axhelm'))"
done
