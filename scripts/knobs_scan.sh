for qg in 8 16; do for sr in 256 512 768 1280; do
  VERS_QG=$qg VERS_SEG_ROWS=$sr python bench.py --steps 6 --warmup 2 --no-cpu --no-recall --kmeans-iters 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('QG=$qg seg=$sr scan ms', d['roofline']['launch_ms'], 'streamed GB', round(d['roofline']['streamed_bytes_per_launch']/1e9,1), 'qps', d['value'], 'items', d['roofline']['work_items'])"
done; done
