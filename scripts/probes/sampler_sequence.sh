R=$PWD; O=$R/gpurun_out/seqs; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace --output-format csv -d $O -- python $R/scripts/bench_workloads.py --only-sampler --steps 6 "$@" > $O/run.log 2>&1
f=$(find $O -name "*kernel_trace.csv" | head -1)
python $R/scripts/probes/step_sequence.py $f > $O/sequence.txt
find $O -name "*kernel_trace*" -delete
