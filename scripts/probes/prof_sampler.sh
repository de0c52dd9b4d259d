R=$PWD; O=$R/gpurun_out/samp; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/scripts/bench_workloads.py --only-sampler --steps 20 > $O/run.log 2>&1
find $O -name "*kernel_trace*" -delete
