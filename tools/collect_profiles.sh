#!/bin/bash
# After `gpurun -- GIT_HEAD=... bash tools/final_verify.sh`: copy the summaries of gpurun_out/final into profiles/rNN (ROUND=r04 by default) and refresh the docs.
set -e
cd "$(dirname "$0")/.."
F=gpurun_out/final; RN=${ROUND:-r06}; P=profiles/$RN
mkdir -p $P
cp "$(ls -t $F/stats/*/*kernel_stats.csv | head -1)" $P/rocprof_kernel_stats_bench.csv     # newest: gpurun merges earlier rounds' files too
cp $F/kernel_by_shape.csv $F/pmc_traffic_bench.json $F/pmc_mfma_bench.json $F/pytest_gpu.txt $F/smoke.txt $P/
for n in default streams2 torchrun_n1 under_rocprof process_group; do [ -f $F/bench_$n.json ] && cp $F/bench_$n.json $P/bench_${RN}_$n.json; done
[ -f $F/bench_all_models.jsonl ] && cp $F/bench_all_models.jsonl $P/bench_${RN}_all_models.jsonl
sed -i '/amdgpu.ids/d' $P/*.txt
python tools/fill_docs.py $F
