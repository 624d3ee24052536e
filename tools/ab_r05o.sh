# runtime switches of the HIP / ROCr layers that change launch or wake-up latency: set in the environment before the process starts
# (round 5: HIP_FORCE_DEV_KERNARG=0 +1.1 ms, =1 = the default; GPU_MAX_HW_QUEUES=8 55.8 ms per step instead of 24.6 (!))
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-zorder ${V:+--voxels $V} 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$name', round(r['ms_per_step'],2), round(r['config']['step_ms_p50'],2))"; }
for rep in 1 2 3; do
run default_$rep X=1
run no_interrupt_$rep HSA_ENABLE_INTERRUPT=0
done
V=20000
for rep in 1 2; do
run default_20k_$rep X=1
run no_interrupt_20k_$rep HSA_ENABLE_INTERRUPT=0
done
