# kernels of one step of a secondary workload in launch order:  bash scripts/probes/workload_sequence.sh {c2|c3|c4|c5|eval|sampler} [precision]
W=$1; P=${2:-bf16}
R=$PWD; O=$R/gpurun_out/seq_${W}_$P; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace --output-format csv -d $O -- python $R/scripts/probes/workload_step.py $W $P 6 > $O/run.log 2>&1
f=$(find $O -name "*kernel_trace.csv" | head -1)
python $R/scripts/probes/step_sequence.py $f $W > $O/sequence.txt
find $O -name "*kernel_trace*" -delete
