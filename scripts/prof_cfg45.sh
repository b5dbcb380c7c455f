# per-kernel tables (timed region) and bench lines of the config-4-shaped and config-5-share runs with the final build
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r03}
prof() { n=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/prof_${T}$n -o bench --output-format csv rocpd -- python3 $R/bench.py "$@" --no-units --no-cpu > $O/prof_${T}$n.log 2>&1 &&
  python3 $R/scripts/prof_window.py $O/prof_${T}$n/bench_results.db $O/prof_${T}$n.log 60 > $O/${T}${n}_timed_region_per_kernel.txt && rm -rf $O/prof_${T}$n
}
prof _cfg4 --lx1 10 --ifheat --no-history --kdim 128 --nel 40,25,20 --steps 2 --warmup 2 &&
prof _cfg5 --lx1 12 --block 4 --no-history --kdim 96 --nel 25,25,40 --steps 2 --warmup 2
rc=$?
cd $R; for n in _cfg4 _cfg5; do grep -h '"metric"' $O/prof_${T}$n.log | cut -c1-200; head -8 $O/${T}${n}_timed_region_per_kernel.txt; done
exit $rc
