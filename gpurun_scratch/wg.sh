cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "large_map or conv3 or strided or linear" 2>&1 | tail -2
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.readline()); rf=r['roofline']
print(round(r['ms_per_step'],2))
for k,v in sorted(rf['all_conv_kernels'].items(), key=lambda kv:-kv[1]['ms'])[:8]: print('   ', k, v['launches'], v['ms'], round(v['tflops'],1))"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/wgp -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
python tools/pmc_summary.py gpurun_out/wgp "wgrad_full_kernel<3, 3>" | tail -1
python tools/pmc_summary.py gpurun_out/wgp "wgrad_full_kernel<2, 4>" | tail -1
rm -rf gpurun_out/wgp
