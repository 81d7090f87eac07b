#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + separate PMC passes of the default bench workload, then the summary.
# usage: tools/profile_bench.sh <tag> [bench.py args, e.g. --set MK2 --batch 1024]
#   -> gpurun_out/<tag>_{trace,fetch,write,sq,lds,f64,tcc}/...  and  gpurun_out/<tag>_summary.md, gpurun_out/<tag>_counters.json
# Afterwards (in the build container): python tools/summarize_profile.py <tag>  copies the judged artefacts into profiles/.
# PMC passes never carry --kernel-trace/--stats or any trace domain (one thing per rocprofv3 run).
TAG=${1:-r02}
shift || true
EXTRA="$@"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline $EXTRA"
python3 $R/tools/kernel_hash.py > $OUT/${TAG}_kernel_hash.txt
rocprofv3 -L > $OUT/${TAG}_counters_available.txt 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- $B --steps 10 --warmup 2 > $OUT/${TAG}_trace.json 2> $OUT/${TAG}_trace.err || { echo "trace pass failed"; exit 1; }
pass() {  # pass <name> <counters...>
    local name=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_$name -- $B --steps 3 --warmup 1 > $OUT/${TAG}_$name.json 2> $OUT/${TAG}_$name.err || echo "$name pass failed"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
pass f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT
pass tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
find $OUT -name "*.csv" -size +8M -delete   # keep the merge under the 64 MiB cap
find $OUT -name "*agent_info.csv" -delete
python3 $R/tools/summarize_profile.py $TAG --here || echo "summary failed"
ls $OUT | grep "^${TAG}_"
