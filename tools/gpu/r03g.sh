cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
export CASES="dense 25841x4640,bcn1_ g0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d gpurun_out/pmc1 -o p -- python tools/bench_split3.py > gpurun_out/r03g_b1.txt 2>&1
python tools/pmc_summary.py $(ls gpurun_out/pmc1/*/p_results.db gpurun_out/pmc1/p_results.db 2>/dev/null | head -1) k_gconv > gpurun_out/r03g_pmc1.txt
rocprofv3 --kernel-trace --pmc TCC_MISS_sum TCC_REQ_sum -d gpurun_out/pmc2 -o p -- python tools/bench_split3.py > gpurun_out/r03g_b2.txt 2>&1
python tools/pmc_summary.py $(ls gpurun_out/pmc2/*/p_results.db gpurun_out/pmc2/p_results.db 2>/dev/null | head -1) k_gconv > gpurun_out/r03g_pmc2.txt
rm -rf gpurun_out/pmc1 gpurun_out/pmc2
python - <<'PY'
import re
for f in ('gpurun_out/r03g_pmc1.txt','gpurun_out/r03g_pmc2.txt'):
    for line in open(f):
        m=re.match(r'void \(anonymous namespace\)::(k_gconv\w*<[^>]*>).*?\s+(\S+)\s+(\d+)\s+([\d.]+)\s*$', line)
        if m: print('%-45s %-26s %4s %16s'%m.groups())
PY
tail -3 gpurun_out/r03g_b1.txt | cut -c1-200
