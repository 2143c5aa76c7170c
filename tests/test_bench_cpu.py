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
    for name in ('mfma_pmc.json', 'pmc_traffic.json'):
        with open(os.path.join(ROOT, 'profiles', name)) as f:
            assert json.load(f).get('stamp') == want, name
