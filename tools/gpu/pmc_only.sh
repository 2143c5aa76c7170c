cd $GRAFT_REPO_ROOT
R=r02z
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc_mfma -o m -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap > /dev/null 2>&1
python tools/pmc_mfma.py $(ls gpurun_out/pmc_mfma/*/m_results.db gpurun_out/pmc_mfma/m_results.db 2>/dev/null | head -1) gpurun_out/${R}_mfma_pmc > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap > /dev/null 2>&1
cp profiles/pmc_traffic.json gpurun_out/${R}_pmc_traffic.json
python tools/pmc_traffic.py $(ls gpurun_out/pmc_fetch/*/f_results.db gpurun_out/pmc_fetch/f_results.db 2>/dev/null | head -1) $(ls gpurun_out/pmc_write/*/w_results.db gpurun_out/pmc_write/w_results.db 2>/dev/null | head -1) gpurun_out/${R}_pmc_traffic.json > gpurun_out/${R}_pmc_traffic.txt
rm -rf gpurun_out/pmc_mfma gpurun_out/pmc_fetch gpurun_out/pmc_write
head -3 gpurun_out/${R}_pmc_traffic.txt; python -c "
import json; d=json.load(open('gpurun_out/${R}_mfma_pmc.json')); print({k:v for k,v in d.items() if k!='kernels'})"
