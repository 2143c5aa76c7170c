set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
(time python -m pytest tests -m gpu -q -x --durations=15) > gpurun_out/r02a_tests.log 2>&1
tail -5 gpurun_out/r02a_tests.log
python bench.py > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err
tail -c 600 gpurun_out/r02a_bench.json
rocprofv3 -L > gpurun_out/counters.txt 2>&1
# PMC pass A: executed MFMA work on the default command's kernels
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc_mfma -o m -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap > gpurun_out/pmc_mfma.log 2>&1
ls gpurun_out/pmc_mfma | head
python tools/pmc_mfma.py $(ls gpurun_out/pmc_mfma/*/m_results.db gpurun_out/pmc_mfma/m_results.db 2>/dev/null | head -1) gpurun_out/r02a_mfma_pmc
# PMC pass B: where the waves of the dominant kernel wait
SHAPES="bcn1_ blur,dense longK" BRIEF=1 REPS=3 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d gpurun_out/pmc_wait -o w -- python tools/bench_gconv.py > gpurun_out/pmc_wait.log 2>&1
python tools/pmc_summary.py $(ls gpurun_out/pmc_wait/*/w_results.db gpurun_out/pmc_wait/w_results.db 2>/dev/null | head -1) k_gconv > gpurun_out/r02a_wait_pmc.txt
cat gpurun_out/r02a_wait_pmc.txt
# tile A/B on the dominant shapes
for t in "" 128x128 64x128 128x128w4; do echo "HPL_TILE=$t"; HPL_TILE=$t python tools/bench_groups.py 2>&1 | grep -E "groups=(1|2) "; HPL_TILE=$t SHAPES="dense longK,bcn1_ 1x1,conv2" BRIEF=1 python tools/bench_gconv.py 2>&1 | tail -3; done > gpurun_out/r02a_tiles.txt 2>&1
cat gpurun_out/r02a_tiles.txt
rm -rf gpurun_out/pmc_mfma/*/*.csv
