"""-m gpu: cx_state_export / cx_state_import.  A restored handle must continue bit for bit where the
exporting handle stood, for every schedule and message family; blobs of another graph or configuration are refused."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L

pytestmark = pytest.mark.gpu


def _all_messages(dev, model, direction):
    return dev.get_messages(model.edge_var, model.edge_fac, direction)


def _same(a, b):
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])


@pytest.mark.parametrize("schedule,materialize", [(L.SCHED_FLOODING, False), (L.SCHED_FUSED, False), (L.SCHED_FUSED, True)])
def test_restored_grid_continues_bit_for_bit(hip_lib, tmp_path, schedule, materialize):
    model = cx.synth.gaussian_grid(37, 23, seed=5)
    a = cx.DeviceGraph(schedule=schedule, materialize_messages_to_factor=materialize)
    cx.synth.load_into_device(model, a, 1e6)
    a.sweep(7)
    path = str(tmp_path / "state.bin")
    a.save_state(path)
    a.sweep(5)
    # a second handle on the same graph, deliberately somewhere else
    b = cx.DeviceGraph(schedule=schedule, materialize_messages_to_factor=materialize)
    cx.synth.load_into_device(model, b, 3.0)
    b.sweep(2)
    b.load_state(path)
    assert b.stats()["sweeps_done"] == 7
    b.sweep(5)
    for direction in (L.TO_VARIABLE, L.TO_FACTOR):
        assert _same(_all_messages(a, model, direction), _all_messages(b, model, direction))
    assert _same(a.get_marginals(model.x_ids), b.get_marginals(model.x_ids))
    assert a.stats()["sweeps_done"] == b.stats()["sweeps_done"] == 12


def test_restore_carries_the_observations(hip_lib):
    """The observed-variable flags and the data live in the blob: a handle that was given OTHER data ends up with the
    exporter's posterior (chain-scan schedule: one sweep = the exact forward-backward result)."""
    ma, mb = cx.synth.ssm_chain(300, seed=1), cx.synth.ssm_chain(300, seed=2)
    a = cx.DeviceGraph(schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_into_device(ma, a)
    blob = a.export_state()          # before any sweep: only the injected data and priors
    a.sweep(1)
    b = cx.DeviceGraph(schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_into_device(mb, b)
    b.sweep(1)
    assert not _same(a.get_marginals(ma.x_ids), b.get_marginals(mb.x_ids))
    b.import_state(blob)
    b.sweep(1)
    assert _same(a.get_marginals(ma.x_ids), b.get_marginals(ma.x_ids))


@pytest.mark.parametrize("d", [2, 4, 64])
def test_restored_multivariate_chain_continues_bit_for_bit(hip_lib, d):
    T = 6
    model = cx.synth.lgssm_chain(T, d=d, seed=3)
    a = cx.DeviceGraph(dim=d, schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, a)
    a.sweep(3)
    blob = a.export_state()
    a.sweep(T)
    b = cx.DeviceGraph(dim=d, schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, b, 10.0)
    b.sweep(1)
    b.import_state(blob)
    b.sweep(T)
    assert _same(a.get_marginals(model.x_ids), b.get_marginals(model.x_ids))
    xe = np.isin(model.edge_var, model.x_ids)
    assert _same(a.get_messages(model.edge_var[xe], model.edge_fac[xe], L.TO_VARIABLE),
                 b.get_messages(model.edge_var[xe], model.edge_fac[xe], L.TO_VARIABLE))


def test_foreign_and_damaged_blobs_are_refused(hip_lib):
    model = cx.synth.gaussian_grid(8, 8, seed=5)
    a = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, a, 1e6)
    a.sweep(3)
    blob = a.export_state()
    before = _all_messages(a, model, L.TO_VARIABLE)
    other = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(cx.synth.gaussian_grid(8, 9, seed=5), other, 1e6)
    with pytest.raises(cx.CortexHipError, match="different graph"):
        other.import_state(blob)
    flood = cx.DeviceGraph(schedule=L.SCHED_FLOODING)
    cx.synth.load_into_device(model, flood, 1e6)
    with pytest.raises(cx.CortexHipError, match="schedule"):
        flood.import_state(blob)
    with pytest.raises(cx.CortexHipError, match="truncated"):
        a.import_state(blob[: blob.size - 100])
    bad = blob.copy(); bad[0] ^= 0xFF
    with pytest.raises(cx.CortexHipError, match="not a state blob"):
        a.import_state(bad)
    with pytest.raises(cx.CortexHipError):
        a.import_state(blob[:10])
    assert _same(before, _all_messages(a, model, L.TO_VARIABLE))     # refused imports leave the handle untouched
    empty = cx.DeviceGraph()
    with pytest.raises(cx.CortexHipError):
        empty.export_state()


def test_blob_of_other_rule_parameters_or_a_doctored_variable_table_is_refused(hip_lib):
    """the fingerprint covers the factor parameters (same topology, other variances: refused) and the
    imported variable table may differ from the handle's only in the observed flags."""
    import dataclasses

    model = cx.synth.gaussian_grid(8, 8, seed=5)
    a = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, a, 1e6)
    a.sweep(2)
    blob = a.export_state()
    other = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(dataclasses.replace(model, factor_var=model.factor_var * 1.5), other, 1e6)
    with pytest.raises(cx.CortexHipError, match="different graph"):
        other.import_state(blob)
    # dim > 1: same chain, other (A, Q)
    m4 = cx.synth.lgssm_chain(6, d=4, seed=3)
    b = cx.DeviceGraph(dim=4)
    cx.synth.load_into_device(m4, b)
    b.sweep(2)
    blob4 = b.export_state()
    c = cx.DeviceGraph(dim=4)
    A, Q = m4.psets[0]
    cx.synth.load_into_device(dataclasses.replace(m4, psets={0: (A, 2.0 * Q), 1: m4.psets[1]}), c)
    with pytest.raises(cx.CortexHipError, match="different graph"):
        c.import_state(blob4)
    # a blob whose variable table carries another degree nibble: refused before anything reaches the device
    hdr = 8 + 4 * 4 + 5 * 8 + 2 * 4 + 8      # StateHeader (cx_api_state.hip), then one 16-byte section header, then vinfo
    bad = blob.copy()
    bad[hdr + 16] = (int(bad[hdr + 16]) & 0xF0) | ((int(bad[hdr + 16]) + 1) & 0x0F)
    before = _all_messages(a, model, L.TO_VARIABLE)
    with pytest.raises(cx.CortexHipError, match="variable table"):
        a.import_state(bad)
    assert _same(before, _all_messages(a, model, L.TO_VARIABLE))


def test_blob_of_an_earlier_state_format_is_refused_with_a_version_error(hip_lib):
    """the fingerprint now covers rule parameters and v2f_stale became a bit-field — blobs written before that carry
    magic "CXSTATE1" and must fail as an old FORMAT, not as "a different graph"."""
    model = cx.synth.ssm_chain(20, seed=3)
    dev = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, dev, seed_variance=1e6)
    dev.sweep(3)
    blob = dev.export_state()
    assert bytes(blob[:8]) == b"CXSTATE2"
    old = blob.copy()
    old[7] = ord("1")
    with pytest.raises(cx.CortexHipError, match="earlier build"):
        dev.import_state(old)
    dev.import_state(blob)
