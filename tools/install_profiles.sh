# copy one measurement set (tools/measure_r06.sh <tag>) from gpurun_out/<tag> into profiles/ as the round files (third argument: prefix, default r06)
# usage: bash tools/install_profiles.sh <tag> "<comment for pmc_traffic.json>"
T=${1:?tag}; C=${2:-"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over python bench.py (round 4)"}
O=gpurun_out/$T
R=${3:-r06}
for f in bench.json bench_steps20_warmup5.json bench_world1_rccl.json soak.json bench_B8.json bench_rotate8.json bench_rotate8_spread20.json bench_ncut.json bench_2rank_gloo_rotate.json bench_2rank_gloo_rotate_plain.json bench_2rank_dry_collectives.json; do [ -s $O/$f ] && cp $O/$f profiles/${R}_$f; done
[ -s $O/kernel_stats.csv ] && cp $O/kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
[ -s $O/kernel_stats_summary.txt ] && cp $O/kernel_stats_summary.txt profiles/${R}_bench_kernel_stats_summary.txt
for f in conv_per_shape.txt hbm_bound_kernels.txt ncut_scenes_in_flight.txt scenes_per_gpu.txt step_vs_scene_size.txt host_vs_device.txt host_threads.txt timeline_sections.txt step_sections.txt stream_overlap.txt stream_queue_probe.txt bn_tile_bench.txt; do [ -s $O/$f ] && cp $O/$f profiles/${R}_$f; done
if [ -s $O/pmc_traffic_raw.txt ]; then
  cp $O/pmc_traffic_raw.txt profiles/${R}_pmc_bench_traffic.txt
  python tools/pmc_to_json.py $O/pmc_traffic_raw.txt profiles/pmc_traffic.json "$C"
fi
