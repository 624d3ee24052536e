cd $GRAFT_REPO_ROOT
python tools/host_threads.py --no-cpu-baseline --no-zorder --steps 20 --warmup 5 2>/dev/null | tail -2
USC3D_PREFETCH_THREAD=0 python tools/host_threads.py --no-cpu-baseline --no-zorder --steps 20 --warmup 5 2>/dev/null | tail -2
bash tools/ab.sh -r 2 plain: ahead30:X=1 2>/dev/null | head -0
for r in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-zorder 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=r['config']; print('plain', round(r['ms_per_step'],2), 'p50', round(c['step_ms_p50'],2))"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-zorder --prefetch-ahead 30 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=r['config']; print('ahead30', round(r['ms_per_step'],2), 'p50', round(c['step_ms_p50'],2))"
done
python tools/host_profile.py --no-cpu-baseline --no-zorder > gpurun_out/r06_host_profile.txt 2>&1; head -90 gpurun_out/r06_host_profile.txt | cut -c1-160
