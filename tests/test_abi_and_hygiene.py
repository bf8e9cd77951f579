"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/cortex_hip.h
declares, fails loudly without a GPU, and nothing in the product reaches into oracle/."""
import ctypes as C
import os
import re

import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "cortex_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cx_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(hip_lib):
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(hip_lib, n), f"{n} declared in include/cortex_hip.h but not exported by libcortex_hip.so"
        assert n in L.SIGNATURES, f"{n} has no ctypes signature in cortex.jl_amd/_lib.py"
    assert sorted(L.SIGNATURES) == names
    assert hip_lib.cx_version() == L.ABI_VERSION


def test_header_constants_match_binding():
    text = open(os.path.join(ROOT, "include", "cortex_hip.h")).read()
    defs = dict(re.findall(r"#define\s+(CX_[A-Z_0-9]+)\s+\(?(-?\d+)\)?", text))
    for k, v in defs.items():
        py = k[3:]
        if hasattr(L, py):
            assert getattr(L, py) == int(v), k
    assert C.sizeof(L.Config) == 32 and C.sizeof(L.Item) == 24


def test_payload_sizes(hip_lib):
    assert hip_lib.cx_payload_doubles(1, L.FORM_MOMENT) == 2
    assert hip_lib.cx_payload_doubles(1, L.FORM_POINT) == 1
    assert hip_lib.cx_payload_doubles(4, L.FORM_MOMENT) == 20
    assert hip_lib.cx_payload_doubles(64, L.FORM_MOMENT) == 4160
    assert hip_lib.cx_payload_doubles(0, L.FORM_MOMENT) == -1


def test_no_cpu_fallback_without_gpu(hip_lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(cx.CortexHipError) as e:
        cx.DeviceGraph()
    assert e.value.code == L.ERR_NO_DEVICE and "no CPU fallback" in e.value.message
    bad = L.Config(4, 0, 1, 0, 1, 0, 0)
    h = C.c_void_p()
    assert hip_lib.cx_create(C.byref(bad), C.byref(h)) == L.ERR_INVALID_ARGUMENT
    assert hip_lib.cx_destroy(None) == L.OK  # idempotent on NULL


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/."""
    offenders = []
    for base in ("cortex.jl_amd", "cortex", "include", "tools"):
        for dirpath, _dirs, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".c", ".sh")):
                    src = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"(from|import)\s+oracle\b|libcortex_oracle|oracle/", src):
                        offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle|import oracle|tests\.helpers", bench)]
    body = bench[bench.index("def cpu_baseline"):bench.index("def main")]
    assert uses and all(bench.index("def cpu_baseline") < u < bench.index("def main") for u in uses), \
        "bench.py may use the checker only inside cpu_baseline()"
    assert "oracle" in body


def test_header_compiles_as_plain_c_and_the_c_client_links(hip_lib, tmp_path):
    """include/cortex_hip.h is a C header (no C++/torch types) and the C client of tests/c links against the library."""
    import subprocess

    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(ROOT, "cortex.jl_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", exe, "-L" + libdir, "-lcortex_hip",
                           "-Wl,-rpath," + libdir])
    import torch
    if not torch.cuda.is_available():
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert out.returncode == 77 and "no CPU fallback" in out.stderr   # fails loudly without a GPU


def test_cpp_host_classes_compile_against_the_abi(tmp_path):
    """include/cortex_hip.hpp (cortex::Handle / HipProcessor / VmpProcessor, SURVEY.md §8b's "equivalent C++ host class")
    builds with g++ -Wall -Wextra -Werror against the C ABI alone and fails loudly without a GPU."""
    import subprocess

    exe = str(tmp_path / "host_class_demo")
    libdir = os.path.join(ROOT, "cortex.jl_amd")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_class_demo.cpp"), "-o", exe, "-L" + libdir, "-lcortex_hip",
                           "-Wl,-rpath," + libdir])
    import torch
    if not torch.cuda.is_available():
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert out.returncode == 77 and "no CPU fallback" in out.stderr


def test_oracle_under_address_and_ub_sanitizers(tmp_path):
    """CPU sanitizers on the checker (GPU ASan is not available on this pool): rebuild oracle/*.c with
    -fsanitize=address,undefined and run a representative slice of the known-answer tests against that build."""
    import subprocess
    import sys

    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=asan, CXO_LIB=os.path.join(ROOT, "oracle", "libcortex_oracle_asan.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    code = ("import sys; sys.path.insert(0, %r); import pytest; "
            "sys.exit(pytest.main(['-x', '-q', '-p', 'no:cacheprovider', %r, '-k', "
            "'oracle and (beta or ssm or tracing or nibble or scan or chain)']))") % (
        ROOT, os.path.join(ROOT, "tests", "test_oracle_reference_kats.py"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr
