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
