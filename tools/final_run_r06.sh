# round-6 closing run on the GPU box: the whole -m gpu suite (timed), then the bench lines that the docs quote
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; mkdir -p $O
s=$(date +%s)
python -m pytest tests -m gpu -q --durations=30 -p no:cacheprovider > $O/tests.log 2>&1; tail -2 $O/tests.log
echo "gpu suite: $(( $(date +%s) - s )) s" | tee $O/suite_seconds.txt
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-zorder 2>/dev/null | grep '^{' | tail -1 > $O/bench_steps20_warmup5.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-zorder --force-dist 2>/dev/null | grep '^{' | tail -1 > $O/bench_world1_rccl.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-zorder --scenes-per-gpu 8 2>/dev/null | grep '^{' | tail -1 > $O/bench_B8.json
for v in 20000 80000 300000; do python bench.py --no-cpu-baseline --no-zorder --voxels $v --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=r['config']; print('voxels=$v', round(r['ms_per_step'],2), 'ms p50', round(c['step_ms_p50'],2))"; done > $O/step_vs_scene_size.txt
python tools/soak.py --steps 400 2>/dev/null | tail -1 > $O/soak.json
python tools/decoder_spans.py --no-cpu-baseline --no-zorder --steps 20 --warmup 5 2>/dev/null | tail -n 12 > $O/step_sections.txt
python tools/host_threads.py --no-cpu-baseline --no-zorder --steps 20 --warmup 5 2>/dev/null | tail -n 1 > $O/host_threads.txt
for f in bench bench_steps20_warmup5 bench_world1_rccl bench_B8; do python -c "import json; r=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); c=r['config']; print('$f', round(r['value'],2), round(r['ms_per_step'],2), 'p50', round(c['step_ms_p50'],2), 'p90', round(c['step_ms_p90'],2), (r.get('roofline') or {}).get('frac_in_step'))"; done
cat $O/step_vs_scene_size.txt $O/host_threads.txt $O/step_sections.txt
