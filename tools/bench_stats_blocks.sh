#!/bin/bash
# scratch: the statistics kernel by grid size (bench record f1, 1024^3 + uint8 mask)
for nb in 1024 2048 4096 8192; do
  echo "== SPC_STATS_BLOCKS=$nb"
  SPC_STATS_BLOCKS=$nb timeout 300 python bench.py --configs-only none --no-cpu-baseline --no-north-star --steps 10 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); r = j['next_rows']['f1_statistics']
        print('   f1_statistics %.4f ms (%.3f)' % (r.get('roofline', r)['kernel_ms'], r.get('roofline', r)['frac']), r.get('verify'))
"
done
