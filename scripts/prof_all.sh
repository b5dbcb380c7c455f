cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02f -o bench --output-format csv rocpd -- python3 $R/bench.py --steps 5 --warmup 3 --no-units --no-cpu > $R/gpurun_out/prof_r02f.log 2>&1 &&
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02f_lx10 -o bench --output-format csv rocpd -- python3 $R/bench.py --lx1 10 --steps 3 --warmup 2 --no-units --no-cpu > $R/gpurun_out/prof_r02f_lx10.log 2>&1 &&
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02f_lx12 -o bench --output-format csv rocpd -- python3 $R/bench.py --lx1 12 --steps 3 --warmup 2 --no-units --no-cpu > $R/gpurun_out/prof_r02f_lx12.log 2>&1 &&
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02f_blk4 -o bench --output-format csv rocpd -- python3 $R/bench.py --block 4 --steps 3 --warmup 2 --no-units --no-cpu > $R/gpurun_out/prof_r02f_blk4.log 2>&1
cd $R; ls gpurun_out/prof_r02f*; tail -c 300 gpurun_out/prof_r02f.log
