bash scripts/ab_b1.sh
VERS_SCAN_DEBUG=16 python bench.py --batch 1 --streams 1 --steps 50 --warmup 20 --no-cpu --no-recall --no-extra --kmeans-iters 2 2>&1 | grep "stamps\] single" | tail -3
