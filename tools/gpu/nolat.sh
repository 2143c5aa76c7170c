cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --steps 200 "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d.get('host_ms_per_step'))"; }
echo "surface: default / no-lattice(single stream) "
run --data surface
run --data surface --no-lattice
echo "surface lattice only (pipeline, no forward):"
python - <<'PY'
import time, types, torch, sys
sys.path.insert(0,'.')
import hplflownet_amd as H
from hplflownet_amd.lattice import LatticePipeline
from hplflownet_amd.synthetic import SCALES_FILTER_MAP, fill_module_, surface_pair, synthetic_pair
dev=torch.device('cuda:0')
margs = types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP, evaluate=True, use_leaky=True, bcn_use_bias=True, bcn_use_norm=True, last_relu=False, DEVICE='cuda')
model = H.HPLFlowNet(margs).to(dev).eval()
gen = H.GenerateDataUnsymmetric(margs, device=dev, wide_up=model.lattice_hint())
for name, mk in (('surface', surface_pair), ('frustum', synthetic_pair)):
    pairs=[]
    for s in range(8):
        p1,p2,_=mk(8192,s); pairs.append((torch.from_numpy(p1.T.copy()).to(dev), torch.from_numpy(p2.T.copy()).to(dev)))
    side=[torch.cuda.Stream(device=dev, priority=-1)]
    for depth in (1,2,4):
        for rep in range(2):
            n=300
            pipe = LatticePipeline(gen, lambda i: pairs[i%8], 0, n, depth=depth, stream=side, for_training=False, native=True, threaded=False)
            torch.cuda.synchronize(); t=time.perf_counter()
            keep=[]
            for _ in range(n):
                tag, lat, ev = pipe.get(); keep.append(lat); keep=keep[-6:]
            torch.cuda.synchronize(); dt=time.perf_counter()-t
        print(name, 'depth', depth, 'lattice builds/s %.0f (%.3f ms)' % (n/dt, dt/n*1e3))
PY
