mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -k "float64" 2>&1 | tail -30 > gpurun_out/r05s6_t.txt
cat gpurun_out/r05s6_t.txt
