# E = 1300 (one rank's share of config 3 on 8 GPUs): A/B timings, kernel trace reduced to the timed region, and SQ / GRBM counters per kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r04}
O=$R/gpurun_out
bash $R/scripts/ab_small.sh > $O/${T}_ab_small.txt 2>&1 &&
cd /tmp &&
rocprofv3 --kernel-trace --stats -d $O/prof_${T}_E1300 -o bench --output-format csv rocpd -- python3 $R/bench.py --nel 13,10,10 --steps 5 --warmup 3 --no-units --no-cpu > $O/prof_${T}_E1300.log 2>&1 &&
python3 $R/scripts/prof_window.py $O/prof_${T}_E1300/bench_results.db $O/prof_${T}_E1300.log 60 > $O/${T}_E1300_timed_region_per_kernel.txt &&
rm -rf $O/prof_${T}_E1300 &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${T}_E1300 -- python3 $R/bench.py --nel 13,10,10 --steps 1 --warmup 1 --no-units --no-cpu > $O/pmc_${T}_E1300.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $O/pmc_${T}_E1300b -- python3 $R/bench.py --nel 13,10,10 --steps 1 --warmup 1 --no-units --no-cpu > $O/pmc_${T}_E1300b.log 2>&1 &&
python3 $R/scripts/pmc_small.py $O/pmc_${T}_E1300 $O/pmc_${T}_E1300b > $O/${T}_E1300_counters.txt 2>&1
rc=$?
rm -rf $O/pmc_${T}_E1300 $O/pmc_${T}_E1300b
cd $R; cat $O/${T}_ab_small.txt; head -40 $O/${T}_E1300_counters.txt
exit $rc
