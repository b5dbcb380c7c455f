# small-mesh kernel variants: parity tests, A/B against NLG_SMALL_E=0 on the same box, then the E = 1300 trace and counters
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_n8.py tests/test_gpu_linop.py tests/test_gpu_fortran.py -x -q 2>&1 | tail -5 &&
echo "--- NLG_SMALL_E=0" && NLG_SMALL_E=0 bash scripts/ab_small.sh && echo "--- default" &&
bash scripts/prof_small.sh ${1:-r04b}
