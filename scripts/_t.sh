cd $GRAFT_REPO_ROOT
( time python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err ) 2> gpurun_out/bench_default.time
tail -3 gpurun_out/bench_default.time
python3 -c "
import json; j=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline']['kernel'], round(j['roofline']['frac'],3), 'gs', round(j['roofline_gs']['frac'],3))
print(json.dumps(j['cpu_baseline'])[:1500])"
