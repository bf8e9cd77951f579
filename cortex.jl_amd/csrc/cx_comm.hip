// cx_comm.hip — the once-per-sweep halo exchange on RCCL, issued from the library (no host language in the loop).
//
// The reference has no distributed path; this is new.  One process per GPU; each rank exchanges the
// variable→factor messages of its cut edges with its partition neighbours: grouped ncclSend/ncclRecv over xGMI on a
// dedicated non-blocking stream, overlapped with the main sweep kernel:
//     main : [v2f of exported slots, pack] --E1--> ............[main sweep kernel]........ --wait E2--> [unpack, push ghosts]
//     comm :                               wait E1 [group{send, recv} x peers] --E2-->
// RCCL is resolved at run time with dlopen (preferring a librccl the process has already loaded, e.g. PyTorch's, so that
// two copies never coexist); the library itself has no link-time dependency on it and still loads on a CPU-only host.

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

#include "cx_internal.h"

namespace cx {

struct NcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static NcclApi g_nccl;

static bool nccl_load(std::string &err) {
    if (g_nccl.lib) return true;
    const char *names[] = {"librccl.so", "librccl.so.1"};
    void *lib = nullptr;
    for (const char *n : names) if (!lib) lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);   // one already in the process
    for (const char *n : names) if (!lib) lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { err = std::string("dlopen(librccl) failed: ") + dlerror(); return false; }
#define CX_SYM(field, name)                                                                 \
    g_nccl.field = (decltype(g_nccl.field))dlsym(lib, name);                                \
    if (!g_nccl.field) { err = std::string("librccl lacks ") + name; return false; }
    CX_SYM(GetUniqueId, "ncclGetUniqueId") CX_SYM(CommInitRank, "ncclCommInitRank") CX_SYM(CommDestroy, "ncclCommDestroy")
    CX_SYM(Send, "ncclSend") CX_SYM(Recv, "ncclRecv") CX_SYM(GroupStart, "ncclGroupStart") CX_SYM(GroupEnd, "ncclGroupEnd")
    CX_SYM(GetErrorString, "ncclGetErrorString")
#undef CX_SYM
    g_nccl.lib = lib;
    return true;
}

bool comm_unique_id(void *out128, std::string &err) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    if (!nccl_load(err)) return false;
    ncclUniqueId id;
    ncclResult_t r = g_nccl.GetUniqueId(&id);
    if (r != ncclSuccess) { err = std::string("ncclGetUniqueId: ") + g_nccl.GetErrorString(r); return false; }
    std::memcpy(out128, &id, 128);
    return true;
}

bool comm_init(cx_handle *h, int world, int rank, const void *id128, std::string &err) {
    if (!nccl_load(err)) return false;
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    ncclComm_t comm = nullptr;
    ncclResult_t r = g_nccl.CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) { err = std::string("ncclCommInitRank: ") + g_nccl.GetErrorString(r); return false; }
    hipError_t e = hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_packed, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_recv, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_swept, hipEventDisableTiming);
    if (e != hipSuccess) { err = std::string("comm stream/events: ") + hipGetErrorString(e); return false; }
    h->comm = comm; h->comm_world = world; h->comm_rank = rank;
    return true;
}

void comm_destroy(cx_handle *h) {
    if (h->comm && g_nccl.CommDestroy) (void)g_nccl.CommDestroy((ncclComm_t)h->comm);
    h->comm = nullptr;
    if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
    if (h->ev_packed) (void)hipEventDestroy(h->ev_packed);
    if (h->ev_recv) (void)hipEventDestroy(h->ev_recv);
    if (h->ev_swept) (void)hipEventDestroy(h->ev_swept);
    h->comm_stream = nullptr; h->ev_packed = h->ev_recv = h->ev_swept = nullptr;
}

// grouped send/recv with every peer, issued on `stream`; returns false with err set on an RCCL error
bool comm_exchange_on(cx_handle *h, hipStream_t stream, std::string &err) {
    if (h->peers.empty()) return true;
    ncclComm_t comm = (ncclComm_t)h->comm;
    ncclResult_t r = g_nccl.GroupStart();
    for (const auto &p : h->peers) {
        const int64_t per = h->cfg.dim == 1 ? 2 : h->nc;     // doubles per message (storage form)
        if (r == ncclSuccess && p.send_count)
            r = g_nccl.Send((const double *)h->d_send_buf + per * p.send_off, (size_t)(per * p.send_count), ncclDouble, p.rank, comm, stream);
        if (r == ncclSuccess && p.recv_count)
            r = g_nccl.Recv((double *)h->d_recv_buf + per * p.recv_off, (size_t)(per * p.recv_count), ncclDouble, p.rank, comm, stream);
    }
    ncclResult_t r2 = g_nccl.GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) { err = std::string("RCCL send/recv: ") + g_nccl.GetErrorString(r); return false; }
    return true;
}

// the per-sweep message halo: on the comm stream, beside the main kernel
bool comm_exchange(cx_handle *h, std::string &err, bool packed_on_comm_stream) {
    if (h->peers.empty()) return true;
    if (!packed_on_comm_stream) {   // the send buffer was packed on the main stream
        (void)hipEventRecord(h->ev_packed, h->stream);
        (void)hipStreamWaitEvent(h->comm_stream, h->ev_packed, 0);
    }
    if (!comm_exchange_on(h, h->comm_stream, err)) return false;
    (void)hipEventRecord(h->ev_recv, h->comm_stream);
    return true;
}

}  // namespace cx
