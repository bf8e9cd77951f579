"""CPU: the host side of the IPC halo transport (cortex.jl_amd/partition.py: DeepHaloIpc) — which of a neighbour's receive segments
each of a rank's send segments is connected to.  A fake device records the calls; what is checked is what bench.py --gpus N relies
on: segment k of mine for rank q lands in q's k-th segment for me, with equal length, and the two hold the SAME (variable, factor)
messages in the same order.  (The device side is covered by tests/test_gpu_halo_ipc.py.)"""
import os

import numpy as np
import pytest

from cortex.jl_amd import partition


class FakeDev:
    dim = 1

    def __init__(self, rank):
        self.rank, self.connected, self.peers = rank, {}, None

    def halo_configure_state(self, sv, sf, rv, rf):
        self.n_send, self.n_recv = len(sv), len(rv)

    def halo_set_layers(self, *a):
        pass

    def halo_peers(self, peers):
        self.peers = list(peers)

    def halo_ipc_alloc(self):
        area = ((self.n_recv * 16 + 4095) // 4096 + 1) * 4096
        return bytes([self.rank]) * 64, 0x1000_0000 * (self.rank + 1), area

    def halo_ipc_set_fused(self, on):
        self.fused = bool(on)

    def halo_ipc_connect(self, peer_index, remote_entry, remote_recv_off, remote_area_bytes, handle=None, same_process_base=None):
        assert (handle is None) != (same_process_base is None)
        self.connected[peer_index] = (remote_entry, remote_recv_off, remote_area_bytes, handle, same_process_base)


@pytest.mark.parametrize("world,rows,cols,depth", [(2, 12, 9, 2), (3, 18, 11, 3), (8, 64, 13, 4)])
def test_send_segments_land_in_the_matching_receive_segments(world, rows, cols, depth):
    parts = [partition.grid_rows_deep(rows, cols, r, world, depth, seed=4) for r in range(world)]
    devs = [FakeDev(r) for r in range(world)]
    exs = [partition.DeepHaloIpc(devs[r], parts[r], connect=False) for r in range(world)]
    infos = {r: exs[r].info for r in range(world)}
    for r in range(world):
        exs[r].connect(infos)
    for r in range(world):
        part = parts[r]
        assert sorted(devs[r].connected) == list(range(len(part.peers))), "every peer entry is connected"
        for i, p in enumerate(part.peers):
            entry, off, area, handle, base = devs[r].connected[i]
            q = parts[p.rank]
            back = q.peers[entry]
            assert back.rank == r and back.recv.start == off and back.recv.stop - back.recv.start == p.send.stop - p.send.start
            assert area == infos[p.rank]["area_bytes"] and (off + p.send.stop - p.send.start) * 16 <= area
            # one process here: connected by device address, never by handle
            assert handle is None and base == infos[p.rank]["base"]
            # the same messages, in the same order
            assert np.array_equal(part.send_var[p.send], q.recv_var[back.recv]) and np.array_equal(part.send_fac[p.send], q.recv_fac[back.recv])


def test_a_rank_that_is_its_own_neighbour_pairs_segment_k_with_segment_k():
    part = partition.deep_self(10, 12, 2, seed=1)
    dev = FakeDev(0)
    ex = partition.DeepHaloIpc(dev, part)                 # world 1: connects at once, by address
    for i, p in enumerate(part.peers):
        entry, off, _area, handle, base = dev.connected[i]
        assert entry == i and off == p.recv.start and handle is None and base == ex.base
        assert np.array_equal(part.send_var[p.send], part.recv_var[p.recv]) and np.array_equal(part.send_fac[p.send], part.recv_fac[p.recv])


def test_another_process_is_connected_by_handle_and_a_length_mismatch_is_refused():
    parts = [partition.grid_rows_deep(12, 9, r, 2, 2, seed=4) for r in range(2)]
    devs = [FakeDev(r) for r in range(2)]
    exs = [partition.DeepHaloIpc(devs[r], parts[r], connect=False) for r in range(2)]
    infos = {r: dict(exs[r].info) for r in range(2)}
    infos[1]["pid"] = os.getpid() + 1                     # rank 1 lives elsewhere
    exs[0].connect(infos)
    _entry, _off, _area, handle, base = devs[0].connected[0]
    assert handle == infos[1]["handle"] and base is None
    bad = {r: dict(infos[r]) for r in range(2)}
    bad[1]["entries"] = [(i, rk, o, n + 1) for i, rk, o, n in bad[1]["entries"]]
    with pytest.raises(ValueError, match="expects"):
        exs[0].connect(bad)
