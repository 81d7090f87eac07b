#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counter> [<counter> ...]   (run on the GPU box; env vars pass through)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$TAG -- python3 $R/tools/time_batch.py 4096 > $OUT/pmc_$TAG.log 2>&1
python3 - $OUT/pmc_$TAG <<'PY'
import csv,glob,sys
from collections import defaultdict
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)
if not f: print("no counter file"); sys.exit()
acc=defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if 'blind_rotate' in r['Kernel_Name']: acc[r['Kernel_Name'][:50]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,c in acc.items():
    for n,v in c.items(): print(k, n, "%.4g"%(sum(v)/len(v)), len(v))
PY
