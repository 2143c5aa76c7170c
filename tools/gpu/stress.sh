# The randomised parity runs behind DESIGN.md section 2 (run on the GPU box from the repo root): lattices (fused / staged builders against
# the C oracle), single layers, whole models (inference and training path), bench-size pairs on more seeds, the dense-surface cloud.
cd $GRAFT_REPO_ROOT
R=${1:-r06}
mkdir -p gpurun_out
{
echo "== stress_lattice --native --cases 200"; python tests/stress/stress_lattice.py --native --cases 200 2>&1 | tail -4
echo "== stress_lattice --cases 60 (staged)"; python tests/stress/stress_lattice.py --cases 60 2>&1 | tail -3
echo "== stress_layers --cases 120"; python tests/stress/stress_layers.py --cases 120 2>&1 | tail -4
echo "== stress_models --cases 24"; python tests/stress/stress_models.py --cases 24 2>&1 | tail -4
echo "== parity_n8192 --seeds 1 2 3 4"; python tests/stress/parity_n8192.py --seeds 1 2 3 4 2>&1 | tail -6
echo "== surface_check"; python tests/stress/surface_check.py 2>&1 | tail -6
} > gpurun_out/${R}_stress.txt 2>&1
tail -40 gpurun_out/${R}_stress.txt
