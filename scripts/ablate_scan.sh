python scripts/bench_flat.py --d 768 --n 200000 --b 8 --iters 20 2>&1 | grep "n="
python scripts/bench_flat.py 2>&1 | grep "n="
for sr in 0 2560 640 320 128; do
  echo "== VERS_SEG_ROWS=$sr"
  if [ $sr = 0 ]; then unset VERS_SEG_ROWS; else export VERS_SEG_ROWS=$sr; fi
  python bench.py --steps 4 --warmup 1 --no-cpu --no-recall --kmeans-iters 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ivf scan ms', d['roofline']['launch_ms'], 'streamed GB', d['roofline']['streamed_bytes_per_launch']/1e9, 'qps', d['value'], 'items', d['roofline']['work_items'])"
done
