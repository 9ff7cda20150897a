#!/bin/bash
# scratch: A/B of two builds of the selection kernels on ONE box (tests/libspcube_hip_old.so against the tree's library)
for round in 1 2; do
  for which in new old; do
    if [ $which = old ]; then cp spectral_cube_amd/libspcube_hip.so /tmp/new.so; cp tests/libspcube_hip_old.so spectral_cube_amd/libspcube_hip.so; fi
    echo "== $which (round $round)"
    timeout 300 python tools/bench_select_offset.py 2>&1 | tail -4
    timeout 300 python tools/bench_select.py 2>&1 | grep -v SELECT_REG | tail -6
    if [ $which = old ]; then cp /tmp/new.so spectral_cube_amd/libspcube_hip.so; fi
  done
done
