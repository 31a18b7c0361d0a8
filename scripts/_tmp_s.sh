python -m pytest tests/test_ivf_gpu.py tests/test_limits_gpu.py tests/test_bench_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -4
python bench.py --batch 1 --streams 1 --steps 300 --warmup 20 --no-cpu --no-recall --no-extra --kmeans-iters 2 2>&1 | tail -1 | cut -c1-400
python bench.py --steps 10 --warmup 3 --no-cpu --kmeans-iters 2 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['extra']['single_query'], j['extra']['reference_mode'])"
