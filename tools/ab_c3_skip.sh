for sd in 1 0 1 0; do
echo "skip_dead=$sd"; SPC_SPECTRAL_SKIP_DEAD=$sd timeout 600 python bench.py --steps 5 --warmup 2 --no-north-star --no-cpu-baseline --configs-only C3 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        L=json.loads(ln)
        for r in L['configs']['C3'][2:]: print('   %-60s %8.3f ms'%(r['name'][:60], r['kernel_ms']))
"
done
