# Regenerates everything under profiles/ in ONE gpurun call:  rm -rf gpurun_out/refresh; gpurun --timeout 1500 -- 'bash scripts/refresh_profiles.sh'
# (delete the local gpurun_out/refresh first: gpurun merges into it and stale rocprofv3 files of another PID would be picked up)
R=$PWD; O=$R/gpurun_out/refresh; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -2 > $O/tests.log
python bench.py > $O/bench_bf16.log 2>&1; tail -1 $O/bench_bf16.log > $O/bench_bf16.json
python bench.py --precision fp32 > $O/bench_fp32.log 2>&1; tail -1 $O/bench_fp32.log > $O/bench_fp32.json
python bench.py --precision bf16x3 --no-cpu-baseline > $O/bench_bf16x3.log 2>&1; tail -1 $O/bench_bf16x3.log > $O/bench_bf16x3.json
python bench.py --precision fp16 --no-cpu-baseline > $O/bench_fp16.log 2>&1; tail -1 $O/bench_fp16.log > $O/bench_fp16.json
python scripts/bench_workloads.py 2>&1 | grep workload > $O/workloads.jsonl
python scripts/bench_workloads.py --precision fp16 2>&1 | grep workload > $O/workloads_fp16.jsonl
python scripts/runner_rate.py 512 4 2>&1 | grep "ms per iteration" > $O/runner_rate.txt
python scripts/runner_rate.py 1400 3 2>&1 | grep "ms per iteration" >> $O/runner_rate.txt
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof > $O/prof_bf16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fp32 -- python $R/bench.py --precision fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-prof > $O/prof_fp32.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof --no-graph > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof --no-graph > $O/pmc_write.log 2>&1
find $O -name "*kernel_trace*" -delete
du -sh $O; find $O -type f | head -30
