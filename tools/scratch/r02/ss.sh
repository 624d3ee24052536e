cd $GRAFT_REPO_ROOT
for sh in 0 3 4 5 6; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --spatial-sort $sh 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.readline()); rf=r['roofline']
print('shift $sh', round(r['ms_per_step'],2), rf['kernel'], round(rf['achieved'],1), round(rf['avg_launch_us'],1))"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ss_$sh -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --spatial-sort $sh > /dev/null 2>&1)
python tools/pmc_summary.py gpurun_out/ss_$sh compact_kernel | grep -A1 "compact_kernel<3>" | tail -1
python tools/pmc_summary.py gpurun_out/ss_$sh "wgrad_full_kernel<3, 3>" | tail -1
rm -rf gpurun_out/ss_$sh
done
