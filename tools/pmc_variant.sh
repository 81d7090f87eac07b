#!/bin/bash
# Developer A/B: PMC counters of the blind-rotate kernel of a variant build.  usage: tools/pmc_variant.sh <tag> <variant>   (on the GPU box)
TAG=$1; VAR=$2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
export THFHE_HIP_LIB=$R/torus-fhe_amd/lib/libthfhe_hip_variants.so THFHE_RING_VARIANT=$VAR
cd /tmp && export TMPDIR=/tmp
pass() { local name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmcv_${TAG}_$name -- python3 $R/tools/time_batch.py 4096 > $OUT/pmcv_${TAG}_$name.log 2>&1 || echo "$name failed"; }
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM
pass lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA
pass f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT
python3 - $OUT $TAG <<'PY'
import csv,glob,sys
from collections import defaultdict
out,tag=sys.argv[1:3]
acc=defaultdict(list)
for f in glob.glob(f"{out}/pmcv_{tag}_*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if 'blind_rotate' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(f"== {tag}")
for n,v in sorted(acc.items()): print(f"{n:28s} {sum(v)/len(v):.5g}")
PY
find $OUT -name "*agent_info.csv" -delete
