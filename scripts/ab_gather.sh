#!/bin/bash
# build_index N=10M k=4096 with the rows placed tile-wise (default) or float4-wise (VERS_GATHER_TILES=0): build seconds + index fingerprint
for v in 0 1 0 1; do echo GATHER_TILES=$v; VERS_GATHER_TILES=$v python bench.py --steps 2 --warmup 1 --no-cpu --no-recall --no-extra 2>&1 | grep -o "build_index (k-means[^;]*;\|index fingerprint[^;]*;" | tr '\n' ' '; echo; done
