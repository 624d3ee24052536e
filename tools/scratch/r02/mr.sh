for i in 1 2 3; do
for ov in 1 0; do
USC3D_OVERLAP_ALLREDUCE=$ov python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600+i*2+ov)) bench.py --gpus 2 --steps 3 --warmup 1 --voxels 40000 --dist-backend gloo --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('ov=$ov', repr(r['config']['loss']), r['config']['grad_allreduce'])
"
done; done
