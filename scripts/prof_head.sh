# kernel-trace profile of the headline bench, reduced to its timed region (scripts/prof_window.py): T = tag of the output file
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-cur}
O=$R/gpurun_out
rocprofv3 --kernel-trace --stats -d $O/prof_$T -o bench --output-format csv rocpd -- python3 $R/bench.py --steps 5 --warmup 3 --no-units --no-cpu > $O/prof_$T.log 2>&1 &&
python3 $R/scripts/prof_window.py $O/prof_$T/bench_results.db $O/prof_$T.log 60 > $O/${T}_timed_region_per_kernel.txt &&
rm -rf $O/prof_$T
head -40 $O/${T}_timed_region_per_kernel.txt
