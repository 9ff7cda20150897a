mkdir -p gpurun_out
timeout 1200 python tools/stress_random.py 150 12 > gpurun_out/r05s6_stress_random.txt 2>&1
tail -40 gpurun_out/r05s6_stress_random.txt
