# Regenerates the round's files under profiles/ in ONE gpurun call:  rm -rf gpurun_out/refresh; gpurun --timeout 3000 -- 'bash scripts/refresh_profiles.sh'
# (delete the local gpurun_out/refresh first: gpurun merges into it and stale rocprofv3 files of another PID would be picked up)
# Every command runs under its own `timeout`: a hung profiler must not take the box with it.
R=$PWD; O=$R/gpurun_out/refresh; rm -rf $O; mkdir -p $O
T="timeout 600"
$T python -m pytest tests -m gpu -x -q > $O/tests_full.log 2>&1; echo "pytest rc=$?" >> $O/tests_full.log; grep -v "^Extension modules" $O/tests_full.log | tail -12 | cut -c1-300 > $O/tests.log
$T python bench.py > $O/bench_bf16.log 2>&1; grep '^{"metric"' $O/bench_bf16.log | tail -1 > $O/bench_bf16.json
$T python bench.py --precision fp16x3 --no-secondary > $O/bench_fp16x3.log 2>&1; grep '^{"metric"' $O/bench_fp16x3.log | tail -1 > $O/bench_fp16x3.json
$T python bench.py --precision fp32 --no-secondary --no-cpu-baseline > $O/bench_fp32.log 2>&1; grep '^{"metric"' $O/bench_fp32.log | tail -1 > $O/bench_fp32.json
$T python bench.py --precision bf16x3 --no-secondary --no-cpu-baseline > $O/bench_bf16x3.log 2>&1; grep '^{"metric"' $O/bench_bf16x3.log | tail -1 > $O/bench_bf16x3.json
$T python bench.py --precision fp16 --no-secondary --no-cpu-baseline > $O/bench_fp16.log 2>&1; grep '^{"metric"' $O/bench_fp16.log | tail -1 > $O/bench_fp16.json
# the RCCL path on the one GPU: world size 1 with a real nccl group; round 6: pack + all-reduce + Adam inside the step's graph
NEAT_FORCE_DIST=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 $T python bench.py --no-secondary --no-cpu-baseline > $O/bench_bf16_rccl_world1.log 2>&1; grep '^{"metric"' $O/bench_bf16_rccl_world1.log | tail -1 > $O/bench_bf16_rccl_world1.json
NEAT_GRAPH_TAIL=0 NEAT_FORCE_DIST=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29612 $T python bench.py --no-secondary --no-cpu-baseline > $O/bench_bf16_rccl_world1_eager_tail.log 2>&1; grep '^{"metric"' $O/bench_bf16_rccl_world1_eager_tail.log | tail -1 > $O/bench_bf16_rccl_world1_eager_tail.json
$T python scripts/bench_workloads.py 2>&1 | grep workload > $O/workloads.jsonl
$T python scripts/bench_workloads.py --precision fp16x3 2>&1 | grep workload > $O/workloads_fp16x3.jsonl
$T python scripts/runner_rate.py 512 4 2>&1 | grep "ms per iteration" > $O/runner_rate.txt
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-prof > $O/prof_bf16.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/fp16x3 -- python $R/bench.py --precision fp16x3 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-prof > $O/prof_fp16x3.log 2>&1
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-prof --no-graph > $O/pmc_fetch.log 2>&1
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-prof --no-graph > $O/pmc_write.log 2>&1
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_x3 -- python $R/bench.py --precision fp16x3 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-prof --no-graph > $O/pmc_fetch_x3.log 2>&1
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_x3 -- python $R/bench.py --precision fp16x3 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-prof --no-graph > $O/pmc_write_x3.log 2>&1
find $O -name "*kernel_trace*" -delete
# SQ counters of the step's kernels (two passes) -> per-kernel table + the counter-derived MFMA utilisation (profiles/mfma_util.json)
cd $R && $T bash scripts/pmc_sq.sh > $O/pmc_sq.log 2>&1; python scripts/pmc_summary.py gpurun_out/pmc2 wsdw layer_kernel_ws wgrad_kernel_h3 dw_gather sdf_adjoint sdf_fused head_chain head_bwd sampler_round > $O/sq_counters.txt 2>&1
python scripts/make_mfma_util.py $(find gpurun_out/pmc2/a -name "*counter_collection.csv" | head -1) bf16 > $O/mfma_util.txt 2>&1
cp profiles/mfma_util.json $O/mfma_util.json
# launch-ordered traces of one replayed step of every workload
for w in "c2 bf16" "c4 bf16" "sampler bf16" "sampler fp16x3" "eval bf16"; do set -- $w; $T bash scripts/probes/workload_sequence.sh $1 $2; cd $R; cp gpurun_out/seq_$1_$2/sequence.txt $O/step_sequence_$1_$2.txt; done
# PMC traffic against the library's algorithmic bytes (fails the refresh beyond +-10 % on the layer class)
cd $R && python scripts/make_traffic.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) bf16 > $O/traffic_bf16.txt 2>&1
python scripts/make_traffic.py $(find $O/pmc_fetch_x3 -name "*counter_collection.csv" | head -1) $(find $O/pmc_write_x3 -name "*counter_collection.csv" | head -1) fp16x3 > $O/traffic_fp16x3.txt 2>&1
cp profiles/traffic.json $O/traffic.json
python scripts/check_traffic.py $O/bench_bf16.json bf16 > $O/traffic_check.txt 2>&1 || echo "TRAFFIC CHECK FAILED" >> $O/traffic_check.txt
cat $O/traffic_check.txt
# the bench line once more with the fresh traffic / utilisation files in place (what profiles/r06_bench_bf16.json is)
$T python bench.py > $O/bench_bf16.log 2>&1; grep '^{"metric"' $O/bench_bf16.log | tail -1 > $O/bench_bf16.json
find $O -name "*counter_collection.csv" -size +20M -delete; du -sh $O
