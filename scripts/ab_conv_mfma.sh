# k_conv3m_scalar (matrix-pipe scalar transport): parity, A/B against NLG_CONV_MFMA=0, the config-4-shaped line single and block of 4
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_heat.py tests/test_gpu_fullsize_configs.py -x -q 2>&1 | tail -4 &&
for v in 0 1; do
  NLG_CONV_MFMA=$v python3 bench.py --nel 40,25,20 --lx1 10 --ifheat --no-history --kdim 128 --steps 3 --warmup 2 --no-cpu --no-units 2>>gpurun_out/ab_conv.err | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('cfg4 MFMA=$v', round(j['value'],3), 'matvecs/s', round(j['ms_per_step'],2), 'ms/step; conv ms/step', r['class_ms_per_step']['conv'], 'launches', r['class_launches_per_step']['conv'])"
done &&
python3 bench.py --nel 40,25,20 --lx1 10 --ifheat --no-history --kdim 128 --block 4 --steps 2 --warmup 1 --no-cpu --no-units 2>>gpurun_out/ab_conv.err > gpurun_out/r04_cfg4_block4.json; python3 -c "import json; j=json.loads(open('gpurun_out/r04_cfg4_block4.json').read().strip().splitlines()[-1]); print('cfg4 block4', round(j['value'],3), 'matvecs/s', round(j['ms_per_step'],2), 'ms per block step', j['config']['launches_per_vector'], 'launches/vector')"
