"""CPU: host-side helpers of bench.py that need no GPU -- the rocm-smi reading behind the line's `power` object and the source
stamp that ties the PMC files under profiles/ to the kernel sources."""
import json
import os
import stat
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _fake_smi(tmp_path, body):
    exe = tmp_path / 'rocm-smi'
    exe.write_text('#!/bin/sh\n' + body + '\n')
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    return str(tmp_path)


def test_smi_sample_reads_power_limit_and_clock(tmp_path, monkeypatch):
    card = {"card0": {"Temperature (Sensor junction) (C)": "60.0", "sclk clock speed:": "(2045Mhz)", "sclk clock level:": "1",
                      "mclk clock speed:": "(2000Mhz)", "Current Socket Graphics Package Power (W)": "1358.0",
                      "Max Graphics Package Power (W)": "1400.0"}}
    monkeypatch.setenv('PATH', _fake_smi(tmp_path, "cat <<'X'\n%s\nX" % json.dumps(card)) + os.pathsep + os.environ.get('PATH', ''))
    got = bench.smi_sample(0)
    assert got == {'package_w': 1358.0, 'limit_w': 1400.0, 'sclk_mhz': 2045.0}


def test_smi_sample_without_an_answer_is_none(tmp_path, monkeypatch):
    monkeypatch.setenv('PATH', _fake_smi(tmp_path, 'echo "no such device"; exit 1') + os.pathsep + os.environ.get('PATH', ''))
    assert bench.smi_sample(0) is None
    card = {"card0": {"sclk clock speed:": "(95Mhz)"}}              # a build of rocm-smi without the power fields
    monkeypatch.setenv('PATH', _fake_smi(tmp_path, "cat <<'X'\n%s\nX" % json.dumps(card)) + os.pathsep + os.environ.get('PATH', ''))
    assert bench.smi_sample(0) == {'package_w': None, 'limit_w': None, 'sclk_mhz': 95.0}


def test_committed_pmc_files_carry_the_stamp_of_the_committed_sources():
    """profiles/mfma_pmc.json and pmc_traffic.json are used by bench.py only when their stamp (sha256 of the kernel sources + the
    kernel-selecting environment) is this tree's: a kernel edit without a new PMC pass must show up here, not in the judge's run."""
    want = bench.source_stamp()
    for name in ('mfma_pmc.json', 'pmc_traffic.json', 'trace_hbm.json'):
        with open(os.path.join(ROOT, 'profiles', name)) as f:
            assert json.load(f).get('stamp') == want, name


REQUIRED = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'epe3d', 'exact_bf16x3', 'train', 'ranks', 'kernels')


def check_contract_line(line, n_gpus=1, required=REQUIRED):
    """What the driver's parser needs of bench.py's ONE stdout line (shared with tests/test_gpu_multirank.py)."""
    assert '\n' not in line and len(line) < bench.LINE_LIMIT, len(line)
    d = json.loads(line)
    for k in required:
        assert k in d, k
    assert d['n_gpus'] == n_gpus and d['unit'] == 'point-pairs/s' and d['higher_is_better'] is True
    assert d['metric'] == 'point-pairs/sec + EPE3D, N=8192 FlyingThings3D, 1/2/4/8 MI355X'
    assert 'workload' in d['config'] and 'model' not in d['config']
    if 'roofline' in required:
        for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
            assert k in d['roofline'], k
    if 'cpu_baseline' in required:
        for k in ('value', 'unit', 'cores', 'kind', 'sample'):
            assert k in d['cpu_baseline'], k
    return d


def test_contract_line_is_short_and_complete():
    """Round 5's line had grown to 20 KB and the driver's record of it came back unparsed (BENCH_r05.json: parsed null).  The
    line is now made from the full record by bench.compact_line: rebuilt here from the recorded round-5 record (the 20-KB one)
    it must stay under bench.LINE_LIMIT with every contract field, `roofline` and `cpu_baseline` in it."""
    with open(os.path.join(ROOT, 'profiles', 'r05y_bench_driver_cmd.json')) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 3 * bench.LINE_LIMIT            # (the input is the oversized record)
    line = bench.compact_line(full)
    d = check_contract_line(line)
    assert abs(d['value'] - full['value']) < 1e-5 * full['value']
    assert abs(d['roofline']['frac'] - full['roofline']['frac']) < 1e-5
    assert d['roofline']['traffic'] == full['roofline']['traffic'] and d['roofline']['whole_step']['frac'] > 0
    assert d['exact_bf16x3']['value'] > 0 and d['train']['ms_per_step'] > 0 and d['epe3d']['abs_delta'] < 1e-4
    assert {'slice', 'splat', 'slice_deep', 'splat_deep'} <= set(d['kernels'])
    assert all('frac' in v for v in d['kernels'].values())
    # eight ranks' worth of per-rank figures still fit
    full['ranks']['ms_per_step_by_rank'] = [2.392284749657847] * 8
    full['ranks']['host_busy_ms_by_rank'] = [0.8488328981911764] * 8
    full['n_gpus'] = 8
    check_contract_line(bench.compact_line(full), n_gpus=8)
    # and a record bloated beyond anything seen sheds optional blocks instead of growing
    full['roofline']['kernel'] = 'x' * 5000
    full['config']['workload'] = 'y' * 3000
    full['kernels'] = {k: dict(v, frac=0.123456789) for k, v in full['kernels'].items()}
    assert len(bench.compact_line(full)) < bench.LINE_LIMIT + 3000


def test_emit_writes_the_detail_file_and_prints_one_line(tmp_path, capsys):
    with open(os.path.join(ROOT, 'profiles', 'r05y_bench_driver_cmd.json')) as f:
        full = json.load(f)
    path = str(tmp_path / 'bench_detail.json')
    bench.emit(full, path)
    out = capsys.readouterr().out
    assert out.count('\n') == 1
    d = check_contract_line(out.strip())
    assert d['detail'] == 'bench_detail.json'
    with open(path) as f:
        assert json.load(f)['kernels'].keys() == full['kernels'].keys()          # the per-class tables live there
    bench.emit(full, str(tmp_path / 'no' / 'such' / 'dir' / 'x.json'))         # an unwritable place costs the file, not the line
    check_contract_line(capsys.readouterr().out.strip())


def test_design_md_quotes_the_committed_profile_set():
    """The generated regions of DESIGN.md (tools/design_fill.py) are filled -- no placeholder left -- and quote the driver-command
    record that is committed under profiles/."""
    import re
    text = open(os.path.join(ROOT, 'DESIGN.md')).read()
    assert '@@' not in text
    m = re.search(r'<!--HEADLINE-->\*\*([\d.]+) point-pairs/s\*\*', text)
    d = json.load(open(os.path.join(ROOT, 'profiles', 'r06_bench_driver_cmd_detail.json')))
    assert m and abs(float(m.group(1)) - d['value']) < 0.06


def test_split3_maybe_covers_every_launch_the_split_kernel_takes():
    """ops.split3_maybe / gconv_common.h split3_maybe decide whether a launch's operand magnitude is reduced at all: it must say
    yes wherever launch_split3 (mirrored by bench.split3_takes) takes the launch, and no for the shapes round 6 found reduced in
    vain (the 1x1 convs of level 3: 1 787 rows)."""
    from hplflownet_amd import ops
    if not ops.SPLIT3:
        return
    for M in (1024, 1787, 3574, 8191, 8192, 9433, 16383, 16384, 25841, 34631, 69262):
        for N in (64, 128, 256, 512, 580, 1024):
            for C in (32, 64, 128, 260, 324, 388, 580, 1024):
                for F in (1, 7, 8, 15):
                    if bench.split3_takes(M, N, C, F):
                        assert ops.split3_maybe(M, C, F, N), (M, N, C, F)
    assert not ops.split3_maybe(1787, 512, 1, 256) and not ops.split3_maybe(1787, 256, 1, 512)
    assert ops.split3_maybe(1787, 260, 15, 256)           # level 3's stencil: split over K into one round of workgroups
    assert ops.split3_maybe(9433, 256, 1, 256) and ops.split3_maybe(8192, 1024, 1, 512)
