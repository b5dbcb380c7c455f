cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_edge_cases.py -x -q -k "deferred or single_reduction" > gpurun_out/t1.log 2>&1; tail -12 gpurun_out/t1.log | cut -c1-300
if grep -q failed gpurun_out/t1.log; then exit 1; fi
python3 -m pytest tests/test_gpu_n8.py tests/test_gpu_linop.py tests/test_gpu_block.py tests/test_gpu_heat.py tests/test_gpu_proj.py tests/test_gpu_known_answer.py -x -q > gpurun_out/t2.log 2>&1; tail -5 gpurun_out/t2.log | cut -c1-300
