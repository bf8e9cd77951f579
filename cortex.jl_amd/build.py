"""Build recipe of libcortex_hip.so (hand-written HIP for gfx950; no hipify, no CUDA shims).

`python -m cortex.jl_amd.build` or `__graft_entry__.build()`; hipcc cross-compiles without a GPU.
The .so is built in-tree so that it travels with the snapshot to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcortex_hip.so")
SOURCES = ["cx_api.hip", "cx_api_mv.hip", "cx_api_msg.hip", "cx_api_sweep.hip", "cx_api_halo.hip", "cx_api_ipc.hip", "cx_api_state.hip", "cx_api_ref.hip", "cx_health.hip", "cx_kernels.hip", "cx_batch.hip", "cx_kary.hip", "cx_kary_mv.hip", "cx_chain.hip", "cx_planscan.hip", "cx_mv.hip", "cx_mvchain.hip", "cx_mv64chain.hip", "cx_mvbatch.hip", "cx_mv64.hip", "cx_mv64w.hip", "cx_comm.hip", "cx_vmp.hip"]
# every header a source may include: a change in any of them rebuilds everything
HEADERS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(ROOT, "include", "cortex_hip.h")]


# which sources a kernel's code comes from: a counter-traffic figure kept under profiles/ is only as good as the kernel it was
# measured on, so the summaries store a hash of these files and bench.py refuses a figure whose kernel has changed since
KERNEL_SOURCES = (("k_sweep_mv", ("cx_mv.hip", "cx_mv_core.h")), ("k_mvc_", ("cx_mvchain.hip", "cx_mv_core.h")),
                  ("k_chain_", ("cx_chain.hip", "cx_lin.h")), ("k_pscan_", ("cx_planscan.hip", "cx_lin.h")), ("k_rule64w", ("cx_mv64w.hip", "cx_mv64w_core.h")), ("k_compose64", ("cx_mv64chain.hip", "cx_mv64w_core.h")), ("k_walk64b", ("cx_mv64chain.hip", "cx_mv64w_core.h")), ("k_rule64", ("cx_mv64.hip",)),
                  ("k_sweep", ("cx_kernels.hip", "cx_scalar_core.h")), ("k_batch", ("cx_batch.hip", "cx_scalar_core.h")), ("k_ref_cluster", ("cx_batch.hip", "cx_scalar_core.h")), ("k_mf_", ("cx_vmp.hip",)), ("k_rate", ("cx_vmp.hip",)), ("k_gamma", ("cx_vmp.hip",)),
                  ("k_set_q", ("cx_vmp.hip",)), ("k_pull", ("cx_vmp.hip",)), ("k_reduce", ("cx_vmp.hip",)))


def sources_sha16(kernel_name: str) -> str:
    """16 hex digits over the sources of `kernel_name`; unknown kernels hash every file of csrc/"""
    import hashlib

    name = kernel_name.replace("cx::", "").replace("void ", "")
    files = None
    for prefix, fs in KERNEL_SOURCES:
        if name.startswith(prefix):
            files = list(fs)          # (cx_internal.h holds the host's handle struct, which changes for reasons no kernel sees)
            break
    if files is None:
        files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    h = hashlib.sha256()
    for f in files:
        h.update(f.encode())
        h.update(_code_only(open(os.path.join(CSRC, f), "rb").read().decode("utf-8", "replace")).encode())
    return h.hexdigest()[:16]


def _code_only(text: str) -> str:
    """the source without comments and blank lines: a profile stays valid over an edit that changes no code"""
    import re

    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    out = []
    for line in text.splitlines():
        i = line.find("//")
        while i >= 0 and line.count('"', 0, i) % 2:      # (inside a string literal: the next one)
            i = line.find("//", i + 2)
        if i >= 0:
            line = line[:i]
        line = line.rstrip()
        if line.strip():
            out.append(line)
    return "\n".join(out)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """One object per source (only stale ones are recompiled, in parallel), then one link."""
    if not force and not is_stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor

    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    dbg = ["-gline-tables-only"] if os.environ.get("CX_BUILD_DEBUG") else []      # line numbers in host backtraces (rocgdb)
    flags = dbg + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
             "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I/opt/rocm/include"]
    hdr_t = max(os.path.getmtime(h) for h in HEADERS)

    def compile_one(src):
        path, obj = os.path.join(CSRC, src), os.path.join(objdir, src + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_t):
            return obj
        extra = ["-DCX_C64_STAMPS=1"] if (src == "cx_mv64chain.hip" and os.environ.get("CX_BUILD_STAMPS")) else []      # lab build: phase stamps
        cmd = [_hipcc()] + flags + extra + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


HOSTLOGIC_SRC = os.path.join(CSRC, "cx_hostlogic.cpp")


def build_hostlogic(asan: bool = False, verbose: bool = False, defines=(), suffix: str = "") -> str:
    """The GPU-free host logic as a CPU library (g++, no HIP) for the CPU tests; `asan`: under -fsanitize=address,undefined;
    `defines` / `suffix`: a variant build (the test that reintroduces a fixed bug to show that the sanitizers catch it)."""
    lib = os.path.join(HERE, "libcortex_hostlogic" + ("_asan" if asan else "") + suffix + ".so")
    deps = [HOSTLOGIC_SRC] + [h for h in HEADERS if os.path.basename(h) in HOSTLOGIC_HEADERS]
    if os.path.exists(lib) and all(os.path.getmtime(d) < os.path.getmtime(lib) for d in deps):
        return lib
    flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"] if asan else ["-O2"]
    cmd = ["g++", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-Werror"] + flags + ["-D" + d for d in defines] + ["-I" + CSRC, "-I" + os.path.join(ROOT, "include"), HOSTLOGIC_SRC, "-o", lib]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return lib


HOSTLOGIC_HEADERS = ("cx_chain64_plan.h", "cx_tree_plan.h", "cx_refsched.h", "cx_flatten.h", "cx_chains.h", "cx_halo_plan.h", "cx_const.h", "cortex_hip.h")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_hostlogic(verbose=True))
