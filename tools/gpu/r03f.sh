cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
HPL_MATH=f32 python tools/dbg_paths.py /tmp/f32.pt
python tools/dbg_paths.py /tmp/s3.pt
python - <<PY
import torch
a=torch.load('/tmp/f32.pt'); b=torch.load('/tmp/s3.pt')
for x in ('native','python'):
    for y in ('native','python'):
        print('f32', x, 'vs split3', y, float((a[x]-b[y]).abs().max()))
PY
