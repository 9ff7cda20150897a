#!/bin/bash
# scratch: A/B of two builds of the moment kernel on ONE box (tests/libspcube_hip_old.so against the tree's library):
# the bench's own C2 and north-star records (signal mask), then the tuning sweep's random mask
for round in 1 2; do
  for which in new old; do
    lib=spectral_cube_amd/libspcube_hip.so
    [ $which = old ] && lib=tests/libspcube_hip_old.so
    echo "== $which (round $round)"
    SPC_HIP_LIBRARY=$PWD/$lib timeout 300 python bench.py --no-configs --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); r = j['roofline']; n = j['north_star']['roofline']
        print('C2 kernel_ms', r['kernel_ms_stats'], 'north star kernel_ms', n.get('kernel_ms_stats', n['kernel_ms']))
"
  done
done
