#!/bin/bash
# usage: keep_profiles.sh TAG ROUND — copies what profiles_run.sh / profiles_extra.sh / live_node.sh left in gpurun_out/ under TAG into
# profiles/ under the round's names (run in the build container after the GPU calls have merged their files back).
T=$1; R=${2:-r3}
cp gpurun_out/${T}_bench_default.json profiles/${R}_bench_default.json
cp gpurun_out/${T}_bench_static.json profiles/${R}_bench_static.json
cp gpurun_out/${T}_bench_minpts0.json profiles/${R}_bench_minpts0.json
cp gpurun_out/${T}_bench_cfg3.json profiles/${R}_bench_cfg3.json
cp gpurun_out/${T}_bench_cfg3_dense.json profiles/${R}_bench_cfg3_dense.json
[ -f gpurun_out/${T}_bench_steps20_warmup5.json ] && cp gpurun_out/${T}_bench_steps20_warmup5.json profiles/${R}_bench_steps20_warmup5.json
for n in shared_bins fixed_grid shared_bins_inflight1 fixed_grid_inflight1 moving_shared_bins moving_fixed_grid; do
  [ -f gpurun_out/${T}_dense_$n.json ] && cp gpurun_out/${T}_dense_$n.json profiles/${R}_cfg3_dense_$n.json
done
for n in cfg2_inflight1 cfg2_inflight3 cfg2_minpts0_inflight1 cfg3_inflight1 cfg3_dense_inflight1; do cp gpurun_out/${T}_${n}_kernel_stats.csv profiles/${R}_${n}_kernel_stats.csv; done
cp gpurun_out/${T}_pmc_cfg2_minpts2_traffic.json profiles/${R}_pmc_traffic_cfg2_minpts2_bucket.json
cp gpurun_out/${T}_pmc_cfg2_minpts0_traffic.json profiles/${R}_pmc_traffic_cfg2_minpts0_bucket.json
cp gpurun_out/${T}_pmc_cfg3_minpts2_traffic.json profiles/${R}_pmc_traffic_cfg3_minpts2_bucket.json
cp gpurun_out/${T}_pmc_cfg3_dense_minpts2_traffic.json profiles/${R}_pmc_traffic_cfg3_dense_minpts2_bucket.json
cp gpurun_out/${T}_sq_counters.txt profiles/${R}_sq_counters.txt
[ -f gpurun_out/${T}_bench_inflight1.json ] && { cp gpurun_out/${T}_bench_inflight1.json profiles/${R}_bench_inflight1.json; cp gpurun_out/${T}_bench_inflight2.json profiles/${R}_bench_inflight2.json
  cp gpurun_out/${T}_bench_jumpevery0.json profiles/${R}_bench_no_box_misses.json; cp gpurun_out/${T}_bench_fixedgrid.json profiles/${R}_bench_fixedgrid_only.json
  cp gpurun_out/${T}_leaf_sweep.txt profiles/${R}_leaf_sweep.txt; cp gpurun_out/${T}_cfg5_dense_one_gpu.json profiles/${R}_cfg5_dense_one_gpu.json; cp gpurun_out/${T}_cfg5_one_gpu.json profiles/${R}_cfg5_one_gpu.json
  cp gpurun_out/${T}_live_node.jsonl profiles/${R}_live_node_after.jsonl; cp gpurun_out/${T}_live_node_kernels.txt profiles/${R}_live_node_kernels_after.txt; }
bash scripts/kernel_resources.sh > profiles/${R}_kernel_resources.txt 2>&1
ls profiles | grep "^${R}_" | wc -l
