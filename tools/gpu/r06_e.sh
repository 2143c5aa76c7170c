cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $O/r06_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/r06_pytest_gpu.txt
bash tools/gpu/profile_round.sh r06
python bench.py --points 500 --steps 60 --warmup 5 --no-cpu-baseline --no-train-probe --detail '' 2>/dev/null | tail -1 | cut -c1-600
