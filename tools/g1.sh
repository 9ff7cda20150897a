mkdir -p gpurun_out
timeout 900 python tools/test_wide_ops.py > gpurun_out/r05s6_wide_ops.txt 2>&1
grep -v "^ok" gpurun_out/r05s6_wide_ops.txt | tail -40; grep -c "^ok" gpurun_out/r05s6_wide_ops.txt
