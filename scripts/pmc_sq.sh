R=$PWD; O=$R/gpurun_out/pmc2; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/a -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-graph --no-secondary > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $O/b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-graph --no-secondary > $O/b.log 2>&1
find $O -name "*kernel_trace*" -delete; du -sh $O; tail -2 $O/a.log
