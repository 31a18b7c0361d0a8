#!/bin/bash
# quick PMC look at the list-scan kernel (k-means cut to 2 iterations; not the judged profile)
set -u
TAG=${1:-q}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu --no-recall --kmeans-iters 2 $*"
pass() { local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$name.log" 2>&1; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
pass fetch FETCH_SIZE
python3 - "$OUT" <<'PY'
import csv,glob,collections,sys
out=sys.argv[1]
for d in sorted(glob.glob(out+'/pmc_*')):
    cnt=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+'/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            if 'IvfSrc' in r['Kernel_Name']:
                cnt[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in cnt.items():
        print(d.split('/')[-1],k)
        for c,vals in sorted(v.items()):
            print('   %-24s %14.5g  (n=%d)'%(c,sum(vals)/len(vals),len(vals)))
PY
