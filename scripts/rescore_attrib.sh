for f in 0 512 1024 1536; do echo "DEBUG=$f"; VERS_SCAN_DEBUG=$f bash scripts/trace_shard.sh 8 2>&1 | grep "ivf_rescore\|kernel time per step"; done
