cd $GRAFT_REPO_ROOT
for c in 1.0 0.7 1.4 2.0 3.0; do
NLG_COARSE_SCALE=$c python3 bench.py --steps 3 --warmup 2 --no-cpu --no-units 2>>gpurun_out/exp.err | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('coarse scale $c', round(j['ms_per_step'],2), 'ms/step; pressure its/step', j['config']['pressure_iters_per_time_step'])"
done
