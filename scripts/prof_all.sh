# final-build evidence for one round: the Fortran GPU tests, kernel-trace profiles of the bench (headline, E=1300, lx1=10, lx1=12,
# block 4) each reduced to its timed region by scripts/prof_window.py, the two PMC passes, the driver's own command line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r04f}
O=$R/gpurun_out
prof() {   # name, bench flags...
  n=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/prof_${T}$n -o bench --output-format csv rocpd -- python3 $R/bench.py "$@" --no-units --no-cpu > $O/prof_${T}$n.log 2>&1 &&
  python3 $R/scripts/prof_window.py $O/prof_${T}$n/bench_results.db $O/prof_${T}$n.log 60 > $O/${T}${n}_timed_region_per_kernel.txt &&
  cp $O/prof_${T}$n/bench_kernel_stats.csv $O/${T}${n}_kernel_stats.csv && rm -rf $O/prof_${T}$n
}
python3 -m pytest $R/tests/test_gpu_fortran.py -q -x > $O/${T}_fortran.log 2>&1 &&
prof "" --steps 5 --warmup 3 &&
prof _E1300 --nel 13,10,10 --steps 5 --warmup 3 &&
prof _lx10 --lx1 10 --steps 3 --warmup 2 &&
prof _lx12 --lx1 12 --steps 3 --warmup 2 &&
prof _blk4 --block 4 --steps 3 --warmup 2 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${T}_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-units --no-cpu > $O/pmc_${T}_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${T}_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-units --no-cpu > $O/pmc_${T}_write.log 2>&1 &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $O/pmc_${T}_mfma -- python3 $R/bench.py --steps 1 --warmup 1 --no-units --no-cpu > $O/pmc_${T}_mfma.log 2>&1 &&
python3 $R/scripts/pmc_kernel_means.py $O/pmc_${T}_mfma k_conv3m > $O/${T}_mfma_counters_k_conv3m.txt 2>&1 &&
python3 $R/scripts/pmc_kernel_means.py $O/pmc_${T}_mfma k_fdm_ext_mfma8 > $O/${T}_mfma_counters_k_fdm_ext_mfma8.txt 2>&1 &&
rm -rf $O/pmc_${T}_mfma &&
cd $R && python3 scripts/pmc_traffic.py gpurun_out/pmc_${T}_fetch gpurun_out/pmc_${T}_write --outdir gpurun_out/${T}_pmc --E 10000 --lx1 8 --dim 3 --mix-from gpurun_out/${T}_timed_region_per_kernel.txt > gpurun_out/${T}_pmc.log 2>&1 &&
rm -rf gpurun_out/pmc_${T}_fetch gpurun_out/pmc_${T}_write &&
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver_cmd.json 2> gpurun_out/${T}_bench_driver_cmd.err
rc=$?
cd $R; tail -c 300 gpurun_out/${T}_fortran.log; tail -c 700 gpurun_out/${T}_bench_driver_cmd.json; du -sh gpurun_out
exit $rc
