#!/bin/bash
# scratch: the new moment kernel's launch parameters at the bench's C2 and north-star records, against the old library, ONE box
run() {
  timeout 300 python bench.py --no-configs --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); r = j['roofline']; n = j['north_star']['roofline']
        print('   C2 %.4f ms (min %.4f)   north star %.3f ms (min %.3f)' % (r['kernel_ms'], r['kernel_ms_stats']['min'], n['kernel_ms'], n['kernel_ms_stats']['min']))
"
}
for v in "4 8 0" "4 8 1" "8 8 0" "8 8 1" "4 4 0" "4 2 0"; do
  set -- $v
  echo "== new ZW=$1 U=$2 XCD=$3"
  SPC_MOMENTS_ZW=$1 SPC_MOMENTS_U=$2 SPC_MOMENTS_XCD=$3 run
done
echo "== old"
SPC_HIP_LIBRARY=$PWD/tests/libspcube_hip_old.so run
