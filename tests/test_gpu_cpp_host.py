"""-m gpu: the C++ host classes of include/cortex_hip.hpp (g++ + the C ABI, nothing else) on the reference's SSM test graphs:
sum-product through cortex::HipProcessor against the exact smoother, structured VMP through cortex::VmpProcessor against
the array form of the reference's update_marginals! (oracle/vmp.py)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import exact, vmp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_classes_drive_the_device(hip_lib, tmp_path):
    exe = str(tmp_path / "host_class_demo")
    libdir = os.path.join(ROOT, "cortex.jl_amd")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_class_demo.cpp"), "-o", exe, "-L" + libdir, "-lcortex_hip",
                           "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rows = {}
    for line in out.stdout.splitlines():
        p = line.split()
        rows.setdefault(p[0], []).append(p[1:])
    y = [2.1, 3.9, 6.2, 8.0, 9.7, 12.3]
    lik = np.array([[float(a), float(b)] for _i, a, b in rows["lik"]])
    assert np.array_equal(lik[:, 0], y) and np.array_equal(lik[:, 1], np.ones(6))     # N(y, 1.0), :424-425
    got = np.array([[float(a), float(b)] for _i, a, b in rows["x"]])
    m, v = exact.ssm_chain_posterior(y, 1.0, 1.0)
    np.testing.assert_allclose(got[:, 0], m, rtol=1e-9)
    np.testing.assert_allclose(got[:, 1], v, rtol=1e-9)
    assert rows["launches"][0][0] == "2"                      # one batch of 6 likelihood messages, one batch of 6 marginals
    assert rows["error"][0][0] == "-2" and "12345" in " ".join(rows["error"][0])
    # structured VMP: 5 x (update x; update [ssnoise, obsnoise])
    yv = [0.05, -0.02, 0.11, 0.23, 0.18, 0.31, 0.27, 0.40]
    arr = vmp.StructuredVMP(yv)
    for _ in range(5):
        arr.update(["x"]); arr.update(["ssnoise", "obsnoise"])
    np.testing.assert_allclose([float(t) for t in rows["ssnoise"][0]], arr.ss, rtol=1e-9)
    np.testing.assert_allclose([float(t) for t in rows["obsnoise"][0]], arr.obs, rtol=1e-9)
    q = np.array([[float(a), float(b)] for _i, a, b in rows["q"]])
    np.testing.assert_allclose(q[:, 0], arr.xm, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(q[:, 1], arr.xw, rtol=1e-9)
