#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the bench lines, latency table, KNN decision, KMS / CCS timings and the 256-party check that profiles/rNN_* record each round.
set -x
O=gpurun_out
for S in SK-80 SK-lib; do python bench.py --set $S --no-cpu-baseline --steps 10 > $O/r03_bench_${S}_n1.json 2> $O/r03_bench_${S}_n1.err; done
python bench.py --set MK2 --batch 1024 --steps 10 > $O/r03_bench_MK2_n1.json 2> $O/r03_bench_MK2_n1.err
python bench.py --set MK4 --batch 1024 --steps 5 > $O/r03_bench_MK4_n1.json 2> $O/r03_bench_MK4_n1.err
python bench.py --set MK4-N2048 --batch 1024 --steps 5 > $O/r03_bench_MK4-N2048_n1.json 2> $O/r03_bench_MK4-N2048_n1.err
python bench.py --set MK4-N2048 --batch 256 --steps 5 --no-cpu-baseline > $O/r03_bench_MK4-N2048_b256_n1.json 2> $O/r03_bench_MK4-N2048_b256_n1.err
python bench.py --set MK5 --batch 1024 --steps 5 --no-cpu-baseline > $O/r03_bench_MK5_n1.json 2> $O/r03_bench_MK5_n1.err
python bench.py --set MK8 --batch 512 --steps 5 --no-cpu-baseline > $O/r03_bench_MK8_n1.json 2> $O/r03_bench_MK8_n1.err
python bench.py --set MK2 --batch 1024 --mode party --steps 5 --no-cpu-baseline > $O/r03_party_MK2_n1.json 2> $O/r03_party_MK2_n1.err
python bench.py --set MK4 --batch 1024 --mode party --steps 5 --no-cpu-baseline > $O/r03_party_MK4_n1.json 2> $O/r03_party_MK4_n1.err
python bench.py --set MK4-N2048 --batch 1024 --mode party --steps 5 --no-cpu-baseline > $O/r03_party_MK4-N2048_n1.json 2> $O/r03_party_MK4-N2048_n1.err
python tools/time_batch.py 1 8 32 256 1024 4096 > $O/r03_time_batch.txt 2>&1
python tools/knn_full_bench.py > $O/r03_knn_full_n1.json 2> $O/r03_knn_full_n1.err
python tools/kms_bench.py KMS2 64 > $O/r03_kms2_b64.json 2>/dev/null
python tools/kms_bench.py KMS2 256 > $O/r03_kms2_b256.json 2>/dev/null
python tools/kms_bench.py KMS2 1024 > $O/r03_kms2_b1024.json 2>/dev/null
python tools/kms_bench.py KMS4 64 > $O/r03_kms4_b64.json 2>/dev/null
python tools/ccs_bench.py > $O/r03_ccs2.json 2>/dev/null
python tools/mk256_check.py 24 64 > $O/r03_mk256_check.json 2> $O/r03_mk256_check.err
echo done
