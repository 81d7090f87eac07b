#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the bench lines, latency table, KNN decision, KMS / CCS timings and the 256-party check that profiles/rNN_* record each round.
set -x
O=gpurun_out
for S in SK-80 SK-lib; do python bench.py --set $S --no-cpu-baseline --steps 10 > $O/r04_bench_${S}_n1.json 2> $O/r04_bench_${S}_n1.err; done
python bench.py --set MK2 --batch 1024 --steps 10 > $O/r04_bench_MK2_n1.json 2> $O/r04_bench_MK2_n1.err
python bench.py --set MK4 --batch 1024 --steps 5 > $O/r04_bench_MK4_n1.json 2> $O/r04_bench_MK4_n1.err
python bench.py --set MK4-N2048 --batch 1024 --steps 5 > $O/r04_bench_MK4-N2048_n1.json 2> $O/r04_bench_MK4-N2048_n1.err
python bench.py --set MK4-N2048 --batch 256 --steps 5 --no-cpu-baseline > $O/r04_bench_MK4-N2048_b256_n1.json 2> $O/r04_bench_MK4-N2048_b256_n1.err
python bench.py --set MK5 --batch 1024 --steps 5 --no-cpu-baseline > $O/r04_bench_MK5_n1.json 2> $O/r04_bench_MK5_n1.err
python bench.py --set MK8 --batch 512 --steps 5 --no-cpu-baseline > $O/r04_bench_MK8_n1.json 2> $O/r04_bench_MK8_n1.err
python bench.py --set MK2 --batch 1024 --mode party --steps 5 --no-cpu-baseline > $O/r04_party_MK2_n1.json 2> $O/r04_party_MK2_n1.err
python bench.py --set MK4 --batch 1024 --mode party --steps 5 --no-cpu-baseline > $O/r04_party_MK4_n1.json 2> $O/r04_party_MK4_n1.err
python bench.py --set MK4-N2048 --batch 1024 --mode party --steps 5 --no-cpu-baseline > $O/r04_party_MK4-N2048_n1.json 2> $O/r04_party_MK4-N2048_n1.err
python tools/time_batch.py 1 8 32 256 512 768 1024 1280 1536 2048 2560 3072 4096 5120 > $O/r04_time_batch.txt 2>&1
python tools/knn_full_bench.py > $O/r04_knn_full_n1.json 2> $O/r04_knn_full_n1.err
for q in 8 64; do python tools/knn_full_bench.py --queries $q > $O/r04_knn_queries_${q}_n1.json 2> $O/r04_knn_queries_${q}_n1.err; done
python tools/kms_bench.py KMS2 64 > $O/r04_kms2_b64.json 2>/dev/null
python tools/kms_bench.py KMS2 256 > $O/r04_kms2_b256.json 2>/dev/null
python tools/kms_bench.py KMS2 1024 > $O/r04_kms2_b1024.json 2>/dev/null
python tools/kms_bench.py KMS4 64 > $O/r04_kms4_b64.json 2>/dev/null
python tools/ccs_bench.py > $O/r04_ccs2.json 2>/dev/null
echo done
