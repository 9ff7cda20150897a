#!/bin/bash
# scratch: bench config records (bash tools/ab_configs.sh C3,C5), tree library against tests/libspcube_hip_old.so, ONE box
W=${1:-C3}
run() {
  timeout 600 python bench.py --configs-only $W --no-cpu-baseline --no-north-star --steps 5 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line)
        for k, recs in j['configs'].items():
            for r in (recs if isinstance(recs, list) else recs.get('records', [])):
                rr = r.get('roofline', r)
                if 'kernel_ms' in rr: print('   %s %-70s %.3f ms (%.3f)' % (k, str(rr.get('kernel', r.get('name')))[:70], rr['kernel_ms'], rr['frac']))
"
}
for round in 1 2; do
  echo "== new"; run
  echo "== old"; SPC_HIP_LIBRARY=$PWD/tests/libspcube_hip_old.so run
done
