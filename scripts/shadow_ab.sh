#!/bin/bash
# fp16 shadow rows (the default) vs f32 rows (VERS_SHADOW=0) on the same index and queries: results must be identical bit for bit
cd "$(dirname "$0")/.."
cmp() { python - "$1" <<'PY'
import numpy as np, sys
a=np.load("/tmp/a.npz"); b=np.load("/tmp/b.npz")
bad=[i for i in range(a["ids"].shape[0]) if not (np.array_equal(a["ids"][i],b["ids"][i]) and np.array_equal(a["dst"][i],b["dst"][i]))]
print(sys.argv[1], "mismatching queries:", len(bad), bad[:5])
PY
}
for cfg in ${CFGS:-"3000000:1024"}; do
  export ROWS=${cfg%%:*} NLIST=${cfg##*:}
  VERS_SHADOW=0 python scripts/shadow_ab.py /tmp/b.npz > /dev/null 2>&1
  VERS_SHADOW=1 python scripts/shadow_ab.py /tmp/a.npz 2>&1 | tail -1
  cmp "rows $ROWS lists $NLIST"
done
