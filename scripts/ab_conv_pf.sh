# k_conv3<8,12> base-flow prefetch: parity, then A/B on one box (NLG_CONV_PF=0 = the round-3 kernel)
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_fortran.py tests/test_gpu_block.py tests/test_gpu_n8.py tests/test_gpu_golden.py -x -q 2>&1 | tail -4 &&
for v in 0 1 0 1; do
  NLG_CONV_PF=$v python3 bench.py --steps 10 --warmup 3 --no-cpu --no-units 2>>gpurun_out/ab_conv.err | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('PF=$v', round(j['ms_per_step'],3), 'conv ms/step', r['class_ms_per_step']['conv'], 'launches', r['class_launches_per_step']['conv'])"
done &&
bash scripts/re40_variants.sh
