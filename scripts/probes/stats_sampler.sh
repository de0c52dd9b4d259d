# per-kernel time of the train step with the ErrorBoundSampler on (host-decided eager, device-decided eager, device-decided HIP graph)
R=$PWD; O=$R/gpurun_out/prof_smp; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -- python $R/scripts/bench_workloads.py --only-sampler --steps 10 > $O/run.log 2>&1
tail -4 $O/run.log
f=$(find $O/s -name "*kernel_stats.csv" | head -1); head -40 $f | cut -c1-220
