// cx_halo_plan.h — the GPU-free parts of the deep halo's set-up (cx_halo_set_layers, cx_halo_ipc_alloc): which slices a sweep has to run
// j sweeps after an exchange, the longest run of slices that hold owned variables only, and inside it the longest run none of whose
// variables WRITES a message of the send list.  Pure host C++ over any struct H with cx_handle's host fields (cx_flatten.h).
#pragma once

#include "cx_flatten.h"

namespace cx {
namespace haloplan {

// lay[v] = distance of variable v from the owned set (0 = owned, depth + 1 = a stand-in beyond the redundant rows).
// trim_lo / trim_hi[L]: first / last slice holding a variable of layer <= L; own_slice_lo / hi: the longest run of owned-only slices.
template <class H>
void layers(H *h, const std::vector<int32_t> &lay, int32_t depth) {
    h->trim_lo.assign(depth + 1, (int32_t)h->nslices); h->trim_hi.assign(depth + 1, -1);
    for (int64_t v = 0; v < h->nv; v++) {
        const int32_t s = (int32_t)(v >> kSliceShift);
        for (int32_t L = std::min<int32_t>(lay[v], depth + 1); L <= depth; L++) {   // a variable of layer l belongs to every set "layer <= L", L >= l
            h->trim_lo[L] = std::min(h->trim_lo[L], s); h->trim_hi[L] = std::max(h->trim_hi[L], s);
        }
    }
    h->own_slice_lo = 1; h->own_slice_hi = 0;
    int run0 = -1;
    for (int64_t sl = 0; sl <= h->nslices; sl++) {
        bool owned = sl < h->nslices;
        for (int64_t v = sl * kBlock; owned && v < std::min<int64_t>(h->nv, (sl + 1) * kBlock); v++) owned = lay[v] == 0;
        if (owned && run0 < 0) run0 = (int)sl;
        if (!owned && run0 >= 0) {
            if ((int)sl - run0 > h->own_slice_hi - h->own_slice_lo + 1) { h->own_slice_lo = run0; h->own_slice_hi = (int)sl - 1; }
            run0 = -1;
        }
    }
    h->halo_depth = depth;
}

// cx_halo_ipc_batch runs the last sweep of a batch in two parts around the push of the next exchange: the message in send-list slot s is
// written by the thread of the variable that owns the partner slot — the quiet run holds no such variable
template <class H>
void quiet_run(H *h) {
    h->ipc_quiet_lo = 1; h->ipc_quiet_hi = 0;
    if (h->own_slice_hi < h->own_slice_lo) return;
    std::vector<int32_t> slot_var(h->nslots, -1);
    for (int64_t e = 0; e < h->ne; e++) slot_var[flat::slot_of_edge_t(h, e)] = h->edge_var[e];
    std::vector<uint8_t> writer(h->nslices, 0);
    for (int32_t sl : h->send_slots) {
        const int32_t p = sl >= 0 && sl < (int32_t)h->nslots ? h->partner[sl] : -1;
        if (p >= 0 && slot_var[p] >= 0) writer[slot_var[p] >> kSliceShift] = 1;
    }
    int run0 = -1;
    for (int sl = h->own_slice_lo; sl <= h->own_slice_hi + 1; sl++) {
        const bool quiet = sl <= h->own_slice_hi && !writer[sl];
        if (quiet && run0 < 0) run0 = sl;
        if (!quiet && run0 >= 0) {
            if (sl - run0 > h->ipc_quiet_hi - h->ipc_quiet_lo + 1) { h->ipc_quiet_lo = run0; h->ipc_quiet_hi = sl - 1; }
            run0 = -1;
        }
    }
}

}  // namespace haloplan
}  // namespace cx
