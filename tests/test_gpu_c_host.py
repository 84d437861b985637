"""GPU: the plain-C host of examples/c_host.c (no Python in the process) runs the prox, the blur operator and
SALSA_v2 through the C-ABI with host pointers and checks its own results."""
import subprocess

import pytest

from test_abi import _build_c_host

pytestmark = pytest.mark.gpu


def test_plain_c_host_runs(tmp_path):
    exe = _build_c_host(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout


def test_plain_c_multi_device_host_runs(tmp_path):
    """examples/c_host_multi.c: one process, sbtv_group with three virtual shards on GPU 0, SALSA_v2_sharded over five
    images, bit-equal to the single-context batch (checked by the program itself)."""
    exe = _build_c_host(tmp_path, "c_host_multi")
    r = subprocess.run([exe, "0", "0", "0"], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout and "group of 3 shard(s)" in r.stdout
