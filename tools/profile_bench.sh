#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + separate PMC passes of the default bench workload.
# usage: tools/profile_bench.sh <tag> [bench.py args, e.g. --set MK2 --batch 1024]   -> gpurun_out/<tag>_{trace,fetch,write,sq,lds,tcc}/...
set -e
TAG=${1:-r01}
shift || true
EXTRA="$@"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- $B --steps 10 --warmup 2 > $OUT/${TAG}_trace.json 2> $OUT/${TAG}_trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -- $B --steps 3 --warmup 1 > $OUT/${TAG}_fetch.json 2> $OUT/${TAG}_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -- $B --steps 3 --warmup 1 > $OUT/${TAG}_write.json 2> $OUT/${TAG}_write.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/${TAG}_sq -- $B --steps 3 --warmup 1 > $OUT/${TAG}_sq.json 2> $OUT/${TAG}_sq.err || echo "sq pass failed"
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/${TAG}_lds -- $B --steps 3 --warmup 1 > $OUT/${TAG}_lds.json 2> $OUT/${TAG}_lds.err || echo "lds pass failed"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${TAG}_tcc -- $B --steps 3 --warmup 1 > $OUT/${TAG}_tcc.json 2> $OUT/${TAG}_tcc.err || echo "tcc pass failed"
find $OUT -name "*.csv" -size +8M -delete   # keep the merge under the 64 MiB cap
ls $OUT
