R=$PWD; O=$R/gpurun_out/fstats; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
for g in 1 2 3; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$g -- python $R/scripts/probes/probe_fused_pmc.py $g save > $O/s$g.log 2>&1
f=$(find $O/s$g -name "*kernel_stats.csv" | head -1); echo "gen $g save:"; grep -i fused $f | cut -d, -f1-4 | cut -c1-200
done
