#!/bin/bash
# scratch: order-statistic tests and timings: former 256-thread table / 512-thread blocks, descriptor loads off / on
mkdir -p gpurun_out/sel
for cfg in "256 0" "512 0" "512 1"; do
  set -- $cfg
  export SPC_SELECT_BT=$1 SPC_SELECT_DESC=$2
  tag=bt$1_desc$2
  python -m pytest tests/test_gpu_ops.py tests/test_gpu_cube.py tests/test_gpu_round2.py tests/test_gpu_round3.py -x -q -k "percentile or median or sigma or mad or select or order or out_of_core" > gpurun_out/sel/tests_$tag.log 2>&1
  echo "== $tag"; tail -2 gpurun_out/sel/tests_$tag.log
  python tools/bench_select.py 2>&1 | grep -v "SPC_SELECT_REG" | tee gpurun_out/sel/bench_$tag.log
done
