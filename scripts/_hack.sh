cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export NLG_HACK_W1=1; else unset NLG_HACK_W1; fi
  python3 bench.py --steps 8 --warmup 2 --no-cpu --no-units 2>>gpurun_out/hack.err | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('E10k w1=$v', round(j['ms_per_step'],3), 'opdiv ms/step', r['class_ms_per_step'].get('opdiv'), r['class_launches_per_step'].get('opdiv'), 'p its', j['config']['pressure_iters_per_time_step'])"
done
