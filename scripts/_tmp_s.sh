python -m pytest tests/test_ivf_gpu.py tests/test_limits_gpu.py tests/test_fuzz_gpu.py tests/test_ivf_cosdist_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -4
VERS_SCAN_DEBUG=16 VERS_SCAN_EVENTS=1 python bench.py --batch 1 --streams 1 --steps 50 --warmup 20 --no-cpu --no-recall --no-extra --kmeans-iters 2 2>&1 | grep "stamps\]  \|stamps\] single" | tail -3
for v in 1 2; do python scripts/bench_host_b1.py 2>&1 | grep -E "pipelined" | head -2 | tr '\n' ' '; echo; done
