# A/B of the convection kernels: parity tests, then the bench with the kernel-trace stats per variant (NLG_CONV_SWEEP)
#   usage: ab_conv.sh TAG "lx1:variant lx1:variant ..."
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-conv}; CASES=${2:-"10:0 10:1 12:0 12:1"}
python3 -m pytest $R/tests/test_gpu_n10.py $R/tests/test_gpu_ops.py -q -x > $O/${T}_tests.log 2>&1 || { tail -30 $O/${T}_tests.log; exit 1; }
for cs in $CASES; do
  lx=${cs%%:*}; sw=${cs##*:}
  export NLG_CONV_SWEEP=$sw
  rocprofv3 --kernel-trace --stats -d $O/prof_${T}_${lx}_$sw -o bench --output-format csv -- python3 $R/bench.py --lx1 $lx --steps 3 --warmup 2 --no-units --no-cpu > $O/${T}_${lx}_$sw.log 2>&1 || { tail -5 $O/${T}_${lx}_$sw.log; exit 1; }
  grep -h "k_conv3" $O/prof_${T}_${lx}_$sw/bench_kernel_stats.csv | cut -c1-60,150-400 > $O/${T}_${lx}_$sw.conv.txt
  rm -rf $O/prof_${T}_${lx}_$sw
done
cd $R; tail -3 $O/${T}_tests.log; for f in $O/${T}_*.conv.txt; do echo $f; cat $f; done; grep -h '"metric"' $O/${T}_*.log | cut -c1-120
