# copy one measurement set (tools/measure_r03.sh <tag>) from gpurun_out/<tag> into profiles/ as the round-3 files
# usage: bash tools/install_profiles.sh <tag> "<comment for pmc_traffic.json>"
T=${1:?tag}; C=${2:-"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over python bench.py (round 3)"}
O=gpurun_out/$T
cp $O/bench.json profiles/r03_bench.json
cp $O/bench_B8.json profiles/r03_bench_B8.json
cp $O/bench_rotate8.json profiles/r03_bench_rotate8.json
cp $O/bench_ncut.json profiles/r03_bench_ncut.json
cp $O/kernel_stats.csv profiles/r03_bench_kernel_stats.csv
cp $O/kernel_stats_summary.txt profiles/r03_bench_kernel_stats_summary.txt
cp $O/conv_per_shape.txt profiles/r03_conv_per_shape.txt
cp $O/hbm_bound_kernels.txt profiles/r03_hbm_bound_kernels.txt
cp $O/ncut_scenes_in_flight.txt profiles/r03_ncut_scenes_in_flight.txt
cp $O/scenes_per_gpu.txt profiles/r03_scenes_per_gpu.txt
cp $O/step_vs_scene_size.txt profiles/r03_step_vs_scene_size.txt
cp $O/pmc_traffic_raw.txt profiles/r03_pmc_bench_traffic.txt
python tools/pmc_to_json.py $O/pmc_traffic_raw.txt profiles/pmc_traffic.json "$C"
