# launch-ordered sampler part of the real step for the one-launch rounds (fused) and the round-5 sequence (NEAT_SAMPLER_UNFUSED=1) on one
# box:  bash scripts/probes/sampler_ab.sh [precision]
P=${1:-bf16}; R=$PWD; mkdir -p $R/gpurun_out/sab; rm -f $R/gpurun_out/sab/all_$P.txt
for v in "unfused NEAT_SAMPLER_UNFUSED=1" "fused X=1"; do
  set -- $v
  env $2 timeout 150 bash scripts/probes/workload_sequence.sh sampler $P
  cd $R
  echo "== $1" >> gpurun_out/sab/all_$P.txt
  grep -n "sampler_\|sdf_fused_w64_kernel<4, true\|points_from_rays\|^launches" gpurun_out/seq_sampler_$P/sequence.txt | cut -c1-110 >> gpurun_out/sab/all_$P.txt
  cp gpurun_out/seq_sampler_$P/sequence.txt gpurun_out/sab/seq_$1_$P.txt
done
