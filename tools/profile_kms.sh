#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + separate PMC passes of tools/kms_bench.py (KMS scheme, mk_gate_nand_new).
# usage: tools/profile_kms.sh <tag> [set] [gates]   -> gpurun_out/<tag>_{trace,sq,lds,f64,fetch}/... ; tools/summarize_kms_profile.py <tag> makes the table
TAG=${1:-r03kms}
SET=${2:-KMS2}
G=${3:-256}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/tools/kms_bench.py $SET $G"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- $B > $OUT/${TAG}_trace.json 2> $OUT/${TAG}_trace.err || { echo "trace pass failed"; exit 1; }
pass() { local name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_$name -- $B > $OUT/${TAG}_$name.json 2> $OUT/${TAG}_$name.err || echo "$name pass failed"; }
pass fetch FETCH_SIZE
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
pass f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT
find $OUT -name "*.csv" -size +8M -delete
find $OUT -name "*agent_info.csv" -delete
ls $OUT | grep "^${TAG}_"
