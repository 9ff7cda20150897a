mkdir -p gpurun_out
timeout 300 tools/micro/vmcnt_pipeline > gpurun_out/r05s6_vmcnt.txt 2>&1
cat gpurun_out/r05s6_vmcnt.txt
