# parity tests of the lx1 > 8 kernels, then bench + per-kernel stats at lx1 = 10 and 12
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-lx}
python3 -m pytest $R/tests/test_gpu_n10.py $R/tests/test_gpu_ops.py -q -x > $O/${T}_tests.log 2>&1 || { tail -30 $O/${T}_tests.log; exit 1; }
for lx in ${2:-10 12}; do
  rocprofv3 --kernel-trace --stats -d $O/prof_${T}_${lx} -o bench --output-format csv -- python3 $R/bench.py --lx1 $lx --steps 3 --warmup 2 --no-units --no-cpu > $O/${T}_${lx}.log 2>&1 || { tail -5 $O/${T}_${lx}.log; exit 1; }
  python3 - $O/prof_${T}_${lx}/bench_kernel_stats.csv > $O/${T}_${lx}.stats.txt <<'PY'
import csv, sys, re
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    n = re.sub(r'\(anonymous namespace\)::', '', r['Name']).split('(')[0].replace('void ', '')
    print('%-44s %6s calls %10.1f us avg %6.2f %%' % (n[:44], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
  rm -rf $O/prof_${T}_${lx}
done
cd $R; tail -2 $O/${T}_tests.log; for lx in ${2:-10 12}; do cat $O/${T}_${lx}.stats.txt; grep -h '"metric"' $O/${T}_${lx}.log | cut -c1-130; done
