cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
export CASES="${CASES:-bcn1_ g0}"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc1 -o p -- python tools/bench_split3.py > gpurun_out/r03d_b1.txt 2>&1
python tools/pmc_summary.py $(ls gpurun_out/pmc1/*/p_results.db gpurun_out/pmc1/p_results.db 2>/dev/null | head -1) k_gconv > gpurun_out/r03d_pmc1.txt
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d gpurun_out/pmc2 -o p -- python tools/bench_split3.py > gpurun_out/r03d_b2.txt 2>&1
python tools/pmc_summary.py $(ls gpurun_out/pmc2/*/p_results.db gpurun_out/pmc2/p_results.db 2>/dev/null | head -1) k_gconv > gpurun_out/r03d_pmc2.txt
rocprofv3 --kernel-trace --pmc TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum -d gpurun_out/pmc3 -o p -- python tools/bench_split3.py > gpurun_out/r03d_b3.txt 2>&1
python tools/pmc_summary.py $(ls gpurun_out/pmc3/*/p_results.db gpurun_out/pmc3/p_results.db 2>/dev/null | head -1) k_gconv > gpurun_out/r03d_pmc3.txt
rm -rf gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3
python - <<'PY'
import re
for f in ('gpurun_out/r03d_pmc1.txt','gpurun_out/r03d_pmc2.txt','gpurun_out/r03d_pmc3.txt'):
    for line in open(f):
        m=re.match(r'void \(anonymous namespace\)::(k_gconv\w*<[^>]*>).*?\s+(\S+)\s+(\d+)\s+([\d.]+)\s*$', line)
        if m and 'gconv3' in m.group(1): print('%-25s %-30s %4s %16s'%m.groups())
PY
tail -2 gpurun_out/r03d_b1.txt | cut -c1-200; tail -3 gpurun_out/r03d_b3.txt | cut -c1-200
