#!/bin/bash
# Same-box A/B of the cache policy of the exact kernels' tile loads (VERS_TILE_AUX: 0 default, 2 nt):
# cfg2 flat scan and the single-query IVF path at cfg3 geometry.
cd "$(dirname "$0")/.."
for aux in 0 2 0 2; do
  VERS_EXTRA_CXXFLAGS=-DVERS_TILE_AUX=$aux python -c "import vers_amd.build as b; b.build(force=True)" > /dev/null 2>&1
  echo "== VERS_TILE_AUX=$aux"
  python scripts/bench_flat.py 2>&1 | grep "scan kernel"
  python bench.py --batch 1 --steps 300 --warmup 20 --no-cpu --no-recall --kmeans-iters 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  b=1: scan us', round(r['launch_ms']*1e3,1), 'frac', r['frac'], 'us/query', round(d['ms_per_step']*1e3,1))"
done
python -c "import vers_amd.build as b; b.build(force=True)" > /dev/null 2>&1
