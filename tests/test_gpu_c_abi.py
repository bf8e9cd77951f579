"""-m gpu: the boundary used from plain C (gcc + include/cortex_hip.h + libcortex_hip.so, nothing else), as a Julia
`ccall` would use it; the printed marginals must equal the exact smoother."""
import os
import subprocess

import numpy as np
import pytest

from oracle import exact

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_drives_the_abi(hip_lib, tmp_path):
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(ROOT, "cortex.jl_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", exe, "-L" + libdir, "-lcortex_hip",
                           "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rows = [l.split() for l in out.stdout.splitlines() if l.startswith("x")]
    got = np.array([[float(r[1]), float(r[2])] for r in rows])
    m, v = exact.ssm_chain_posterior([2.1, 3.9, 6.2, 8.0, 9.7], 1.0, 1.0)
    np.testing.assert_allclose(got[:, 0], m, rtol=1e-12)
    np.testing.assert_allclose(got[:, 1], v, rtol=1e-12)
    assert "unknown-edge status -2" in out.stdout and "12345" in out.stdout
    (health,) = [l.split()[1:] for l in out.stdout.splitlines() if l.startswith("health ")]
    assert int(health[0]) > 0 and [int(x) for x in health[1:]] == [0, 0, 0]      # cx_message_health: every message into a state is defined, none is broken
    # the dim = 4 chain driven batch by batch in the reference's schedule (abi_smoke.c: mv_batches)
    d = 4
    rows4 = np.array([[float(x) for x in l.split()[2:]] for l in out.stdout.splitlines() if l.startswith("m4 ")])
    assert rows4.shape == (3, d + d * d)
    A = 0.9 * np.eye(d) + 0.1 * np.eye(d, k=1)
    y = np.array([[0.5, -1.0, 2.0, 0.25], [1.5, -0.5, 1.0, 0.75], [2.5, 0.5, 0.0, 1.25]])
    em, ecov = exact.lgssm_posterior(y, A, 0.1 * np.eye(d), np.eye(d))
    np.testing.assert_allclose(rows4[:, :d], em, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(rows4[:, d:].reshape(3, d, d), ecov, rtol=1e-10, atol=1e-12)
