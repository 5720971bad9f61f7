#!/bin/bash
# Round-end verification on the GPU box: full GPU suite, kernel-trace stats, PMC passes (HBM traffic, MFMA busy), bench lines.
#   GIT_HEAD=<commit> bash tools/final_verify.sh [quick]     (the box has no .git: the caller passes HEAD)
set -u
ROOTD=${GRAFT_REPO_ROOT:-$PWD}
R=$ROOTD/gpurun_out/final
STEPS_PROF=20
rm -rf $R; mkdir -p $R
cd $ROOTD
python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > $R/pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats -- python3 $ROOTD/bench.py --steps $STEPS_PROF --warmup 3 --no-cpu-baseline --no-in-flight --no-alt-splits --no-other-configs > $R/bench_under_rocprof.json 2> $R/rocprof_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/pmc_fetch -- python3 $ROOTD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-in-flight --no-alt-splits --no-other-configs > /dev/null 2> $R/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/pmc_write -- python3 $ROOTD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-in-flight --no-alt-splits --no-other-configs > /dev/null 2> $R/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/pmc_mfma -- python3 $ROOTD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-in-flight --no-alt-splits --no-other-configs > /dev/null 2> $R/pmc_mfma.err
cd $ROOTD
python tools/trace_by_shape.py $R/stats $R/kernel_by_shape.csv 26
python tools/pmc_traffic.py $R/pmc_fetch $R/pmc_write $R/pmc_traffic_bench.json "${GIT_HEAD:-unknown}"
python tools/pmc_mfma.py $R/pmc_mfma $R/pmc_mfma_bench.json
mkdir -p profiles/${ROUND:-r06} && cp $R/pmc_traffic_bench.json profiles/${ROUND:-r06}/pmc_traffic_bench.json   # so that the bench line below reads THIS build's traffic
python bench.py > $R/bench_default.json 2> $R/bench_default.err
if [ "${1:-}" != "quick" ]; then
  for m in openvis_online san_online brivis brivis_swinl openvis_swinl; do python bench.py --model $m --steps 10 --warmup 2 2>> $R/bench_models.err | tail -1 >> $R/bench_all_models.jsonl; done
  python bench.py --streams 2 --steps 32 --warmup 4 --no-cpu-baseline > $R/bench_streams2.json 2>> $R/bench_models.err
fi
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $R/bench_torchrun_n1.json 2> $R/bench_torchrun_n1.err
python bench.py --process-group --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-alt-splits --no-in-flight --gather-masks > $R/bench_process_group.json 2> $R/bench_process_group.err   # split_clip over a ONE-rank RCCL group
python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.txt 2>&1
# keep only the summaries (the raw traces are large)
find $R -name "*kernel_trace.csv" -delete; find $R -name "*agent_info.csv" -delete; find $R -name "*counter_collection.csv" -delete
ls -la $R $R/stats/* | head -40
cat $R/pytest_gpu.txt; tail -c 900 $R/bench_default.json; cat $R/smoke.txt | tail -2; tail -c 300 $R/bench_torchrun_n1.json
