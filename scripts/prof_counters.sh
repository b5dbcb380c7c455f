# SQ counters per kernel at the headline size: what bounds each kernel besides HBM (wave lifetime, wait share, VALU issue, LDS bank conflicts)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r04}
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES --output-format csv -d $O/pmc_${T}_a -- python3 $R/bench.py --steps 1 --warmup 1 --no-units --no-cpu > $O/pmc_${T}_a.log 2>&1 &&
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_${T}_b -- python3 $R/bench.py --steps 1 --warmup 1 --no-units --no-cpu > $O/pmc_${T}_b.log 2>&1 &&
python3 $R/scripts/pmc_small.py $O/pmc_${T}_a $O/pmc_${T}_b > $O/${T}_E10k_counters.txt 2>&1
rc=$?
rm -rf $O/pmc_${T}_a $O/pmc_${T}_b
head -30 $O/${T}_E10k_counters.txt | cut -c1-230
exit $rc
