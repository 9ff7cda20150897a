#!/bin/bash
# scratch: the bench's f1 / f4 records (statistics, median, sigma clip at 1024^3 + uint8 mask), tree library against tests/libspcube_hip_old.so, ONE box
run() {
  timeout 300 python bench.py --configs-only none --no-cpu-baseline --no-north-star --steps 10 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line)
        print('   C2 %.4f ms' % j['roofline']['kernel_ms'], ' | '.join('%s %.4f ms (%.3f)' % (k, r.get('roofline', r)['kernel_ms'], r.get('roofline', r)['frac']) for k, r in j['next_rows'].items() if isinstance(r, dict) and 'kernel_ms' in r.get('roofline', r)))
"
}
for round in 1 2; do
  echo "== new"; run
  echo "== old"; SPC_HIP_LIBRARY=$PWD/tests/libspcube_hip_old.so run
done
