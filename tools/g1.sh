mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|^FAILED|assert" | tail -8 > gpurun_out/r05s6_t.txt
cat gpurun_out/r05s6_t.txt
