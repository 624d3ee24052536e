cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./tools/probes/mfma_rate > gpurun_out/r06_mfma_rate.txt 2>&1; cat gpurun_out/r06_mfma_rate.txt
python tools/decoder_spans.py --no-cpu-baseline --no-zorder --steps 20 --warmup 5 > gpurun_out/r06_step_sections.txt 2>&1; cat gpurun_out/r06_step_sections.txt
python tools/host_vs_device.py --no-cpu-baseline --no-zorder > gpurun_out/r06_host_vs_device.txt 2>&1; tail -4 gpurun_out/r06_host_vs_device.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-zorder 2>/dev/null | tail -1 > gpurun_out/r06_bench_steps20_warmup5.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-zorder --force-dist 2>/dev/null | tail -1 > gpurun_out/r06_bench_world1_rccl.json
for f in r06_bench_steps20_warmup5 r06_bench_world1_rccl; do python -c "import sys,json; r=json.loads(open('gpurun_out/$f.json').read()); c=r['config']; print('$f', round(r['ms_per_step'],2), 'p50', round(c['step_ms_p50'],2), 'p90', round(c['step_ms_p90'],2), c.get('grad_allreduce'))"; done
