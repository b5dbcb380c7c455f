# E = 1300 proxy and headline, three runs each (same box)
cd $GRAFT_REPO_ROOT
for k in 1 2 3; do python3 bench.py --nel 13,10,10 --steps 20 --warmup 5 --no-cpu --no-units 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('E1300', round(j['ms_per_step'],3))"; done
for k in 1 2; do python3 bench.py --steps 10 --warmup 3 --no-cpu --no-units 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('E10k', round(j['ms_per_step'],3), j['roofline']['avg_ms'])"; done
