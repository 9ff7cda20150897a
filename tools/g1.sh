mkdir -p gpurun_out
timeout 900 python tools/test_bilinear_lerp.py time > gpurun_out/r05s6_bilinear_lerp2.txt 2>&1
tail -15 gpurun_out/r05s6_bilinear_lerp2.txt
