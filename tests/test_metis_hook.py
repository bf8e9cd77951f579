"""CPU: the METIS branch of partition.auto_partition (BASELINE config 4 names a "METIS 8-way cut").

libmetis is not in this image, so the ctypes call of partition.metis_assignment is executed here against a stand-in
shared object compiled in the test: it exports METIS_PartGraphKway with METIS 5's C signature, CHECKS what it is handed
(CSR of a symmetric graph without self loops, ncon = 1, null weight / option arrays) and returns a fixed, non-contiguous
assignment.  What is tested is the product's side of the call — argument marshalling, the use of the returned parts — and
that a partition built from that assignment still reproduces the single-process sweep bit for bit."""
import ctypes
import ctypes.util
import os
import subprocess

import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import partition
from tests.helpers import flood_oracle_from_model

STUB = r"""
#include <stdint.h>
#include <stddef.h>
typedef int32_t idx_t; typedef float real_t;
static idx_t last_nvtxs = -1, last_nparts = -1, calls = 0;
idx_t stub_last_nvtxs(void) { return last_nvtxs; }
idx_t stub_last_nparts(void) { return last_nparts; }
idx_t stub_calls(void) { return calls; }
/* METIS 5 API: METIS_OK = 1, METIS_ERROR_INPUT = -2 */
int METIS_PartGraphKway(idx_t *nvtxs, idx_t *ncon, idx_t *xadj, idx_t *adjncy, idx_t *vwgt, idx_t *vsize, idx_t *adjwgt,
                        idx_t *nparts, real_t *tpwgts, real_t *ubvec, idx_t *options, idx_t *objval, idx_t *part) {
    calls++;
    if (!nvtxs || !ncon || !xadj || !adjncy || !nparts || !objval || !part) return -2;
    if (*ncon != 1 || vwgt || vsize || adjwgt || tpwgts || ubvec || options) return -2;
    const idx_t n = *nvtxs;
    if (n <= 0 || *nparts < 1 || xadj[0] != 0) return -2;
    for (idx_t i = 0; i < n; i++) {
        if (xadj[i + 1] < xadj[i]) return -2;
        for (idx_t k = xadj[i]; k < xadj[i + 1]; k++) {
            const idx_t j = adjncy[k];
            if (j < 0 || j >= n || j == i) return -2;
            int back = 0;                                    /* symmetric */
            for (idx_t m = xadj[j]; m < xadj[j + 1]; m++) back |= adjncy[m] == i;
            if (!back) return -2;
        }
    }
    last_nvtxs = n; last_nparts = *nparts;
    /* a cut no contiguous-block fallback would produce: 2 x 2 checkerboard blocks of the vertex order */
    for (idx_t i = 0; i < n; i++) part[i] = ((i / 3) + (i / 11)) % *nparts;
    *objval = 0;
    return 1;
}
"""


@pytest.fixture()
def metis_stub(tmp_path, monkeypatch):
    src, so = tmp_path / "metis_stub.c", tmp_path / "libmetis_stub.so"
    src.write_text(STUB)
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", str(so), str(src)])
    real = ctypes.util.find_library
    monkeypatch.setattr(ctypes.util, "find_library", lambda name: str(so) if name == "metis" else real(name))
    return ctypes.CDLL(str(so))


@pytest.mark.parametrize("world,depth", [(3, 0), (2, 2)])
def test_auto_partition_calls_metis_and_uses_its_parts(metis_stub, world, depth):
    from tests._dist_worker import OracleStateSweeper, OracleSweeper

    whole = cx.synth.gaussian_grid(7, 6, seed=13)
    got = partition.metis_assignment(whole, world)
    assert got is not None, "the k-way branch must accept a loadable libmetis"
    vids, part = got
    n_vars = len(np.unique(whole.edge_var))
    assert metis_stub.stub_last_nvtxs() == n_vars and metis_stub.stub_last_nparts() == world
    assert np.array_equal(part, ((np.arange(n_vars) // 3) + (np.arange(n_vars) // 11)) % world)
    calls0 = metis_stub.stub_calls()
    parts = [partition.auto_partition(whole, r, world, depth=depth) for r in range(world)]
    assert metis_stub.stub_calls() == calls0 + world             # every rank asked METIS, none fell back
    own = [p.model.x_ids if p.owned_x is None else p.owned_x for p in parts]
    assert np.array_equal(np.sort(np.concatenate(own)), np.sort(whole.x_ids))
    for r, o in enumerate(own):                                   # ownership is METIS's answer, not contiguous blocks
        assert np.array_equal(np.sort(o), np.sort(whole.x_ids[part[np.searchsorted(vids, whole.x_ids)] == r]))
    # the cut reproduces the un-partitioned flooding sweeps bit for bit (in-process exchange, CPU checker as the sweeper)
    sweeps = 7
    if depth:
        sws = [OracleStateSweeper(p, 1e6) for p in parts]
        for k in range(sweeps):
            if k % depth == 0:
                for sw in sws:
                    sw.pack()
                for r, p in enumerate(parts):
                    for peer in p.peers:
                        back = [pp for pp in parts[peer.rank].peers if pp.rank == r][0]
                        sws[peer.rank].recv[back.recv] = sws[r].send[peer.send]
                for sw in sws:
                    sw.unpack()
            for sw in sws:
                sw.sweep()
    else:
        sws = [OracleSweeper(p, 1e6) for p in parts]
        for _ in range(sweeps):
            for sw in sws:
                sw.sweep_begin()
            for r, p in enumerate(parts):
                for peer in p.peers:
                    back = [pp for pp in parts[peer.rank].peers if pp.rank == r][0]
                    sws[peer.rank].recv[back.recv] = sws[r].send[peer.send]
            for sw in sws:
                sw.sweep_main(); sw.sweep_end()
    g = flood_oracle_from_model(whole, 1e6)
    g.sweep(sweeps)
    gm, gv = g.marginals()
    for o, sw in zip(own, sws):
        m, v = sw.g.marginals()
        li, wi = np.searchsorted(sw.g.var_ids, o), np.searchsorted(g.var_ids, o)
        assert np.array_equal(m[li], gm[wi], equal_nan=True) and np.array_equal(v[li], gv[wi], equal_nan=True)


def test_metis_error_status_falls_back_to_blocks(metis_stub, monkeypatch):
    """a libmetis that rejects the call (status != METIS_OK): auto_partition cuts contiguous blocks instead"""
    whole = cx.synth.gaussian_grid(5, 4, seed=1)
    # world = 0 parts is refused by the stub with METIS_ERROR_INPUT
    assert partition.metis_assignment(whole, 0) is None
