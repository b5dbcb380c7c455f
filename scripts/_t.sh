cd $GRAFT_REPO_ROOT
for v in 4 3 6 4 3 6; do
NLG_AXHELM_WPB=$v python3 bench.py --steps 8 --warmup 2 --no-cpu --no-units 2>>gpurun_out/hack.err | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('E10k wpb=$v', round(j['ms_per_step'],3), 'axhelm ms/step', r['class_ms_per_step'].get('axhelm'), r['class_launches_per_step'].get('axhelm'))"
done
