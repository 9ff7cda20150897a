mkdir -p gpurun_out
( timeout 1500 python tools/stress_random.py 400 21 2>&1 | grep -v "^(unsupported)" | tail -15
  timeout 900 python tools/stress_random.py 300 22 2>&1 | grep -v "^(unsupported)" | tail -8
  timeout 600 python tools/test_bilinear_lerp.py 2>&1 | grep -v "^ok" | tail -5
  timeout 600 python tools/stress_determinism.py 2>&1 | tail -8 ) > gpurun_out/r05s6_long_fuzz.txt 2>&1
tail -40 gpurun_out/r05s6_long_fuzz.txt
