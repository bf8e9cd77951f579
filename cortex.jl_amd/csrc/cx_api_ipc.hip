// cx_api_ipc.hip — deep-halo state exchange WITHOUT a collective library: every rank pushes the state of its boundary rows
// straight into a receive area of its neighbour (device memory the neighbour exported with hipIpcGetMemHandle, mapped over xGMI)
// and raises an epoch flag there; the neighbour's unpack kernel waits for the flag and scatters the area into its redundant rows.
// Two launches on the handle's own stream per exchange — push, wait + unpack — against pack, RCCL send/recv kernel, unpack of
// cx_halo_state_exchange (cx_api_halo.hip), which stays the audited fallback.  (SURVEY.md §5 names this form as the alternative;
// the reference itself is single-process: the partition is this library's, docs/src/index.md "not implemented: parallel".)
//
// Memory.  One block per rank, allocated fine-grained (every access coherent between agents while kernels run):
//     [ flags: 64 B per peer entry ]  [ receive area, parity 0 ]  [ receive area, parity 1 ]
// A receive area has the layout of the handle's halo receive buffer (segment of peer entry i at recv_off[i] messages).  Exchange
// e (1, 2, ...) uses parity e & 1.  Why two parities suffice without an acknowledgement: a neighbour can only push exchange e + 2
// after it has unpacked exchange e + 1 from me, which I push after my sweeps of batch e + 1, which follow my unpack of exchange e
// on my stream.
// Ordering.  push: every thread stores its messages (system scope, write-through), waits for the acknowledgements, the workgroup
// counts itself done; the LAST workgroup stores the epoch into the neighbours' flags.  unpack: one thread per workgroup polls the
// rank's own flags (system scope, past the caches), bounded by a wall-clock limit — a neighbour that never arrives sets the
// error word that cx_halo_ipc_status reports, the grid drains either way; then the area is read past the caches.

#include "cx_host.h"
#include "cx_halo_plan.h"

using namespace cxh;

namespace {

constexpr int kMaxPeers = 8;
constexpr int64_t kFlagStride = 64;              // bytes between two flags: one per cache line
constexpr int64_t kFlagBytes = 4096;

struct PushArgs {
    int n_peers;
    int64_t send_off[kMaxPeers], send_count[kMaxPeers];
    double2 *remote_area[kMaxPeers];             // the neighbour's receive area of this exchange's parity, at my segment
    unsigned long long *remote_flag[kMaxPeers];
};

struct WaitArgs {
    int n_peers;
    const unsigned long long *flag[kMaxPeers];
};

// Accesses to the exchanged memory carry their own scope (relaxed atomics at system scope: write-through stores, cache-bypassing
// loads) instead of release / acquire FENCES: a fence at agent or system scope writes back / invalidates the whole L2 of the XCD,
// once per wave that executes it — measured: + 100 us per exchange and the sweeps that follow start from a cold L2.  What a
// fence would order is ordered here by completion: a wave waits until its stores have been acknowledged (s_waitcnt vmcnt(0)),
// then the workgroup counts itself done; the flags are stored after the last count.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_sys(double2 *p, double2 v) {      // one 16-byte write-through store (sc0 sc1 = system scope)
    u32x4 r;
    __builtin_memcpy(&r, &v, 16);
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(r) : "memory");
}
// four 16-byte loads past the caches, in flight together; the wait is part of the statement (the compiler does not know that an
// asm load completes later and would read its destination at once)
__device__ __forceinline__ void load_sys4(const double2 *p0, const double2 *p1, const double2 *p2, const double2 *p3, u32x4 (&r)[4]) {
    asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\t"
                 "global_load_dwordx4 %1, %5, off sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc0 sc1\n\t"
                 "global_load_dwordx4 %3, %7, off sc0 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
                 : "memory");
}

constexpr int kIpcBlock = 1024, kIpcItems = 4;   // few, fat workgroups: the completion counter is one same-address atomic per workgroup
                                                 // (884 of them cost more than the copy itself: 14 us per push)

// What travels is 16-byte UNITS: a scalar message is one (its double2), a message of dim 2 - 4 is ncp of them — the pairs of its
// block-major storage form (cx_mv_core.h: slot_offset; pair q of slot s is unit ((s >> 8) ncp + q) 256 + (s & 255) of the buffer).
// Unit u of a halo list is pair u % ncp of its message u / ncp; send / receive offsets and counts are in units.
// (round 6) the matrix-core dims (16, 32, 64) store a message as one record of nc doubles: unit q of slot s is unit s nc / 2 + q; the
// launchers say so with a NEGATIVE unit count
__device__ __forceinline__ int64_t ipc_unit(int32_t slot, int q, int ncp) {
    return ncp < 0 ? (int64_t)slot * (-ncp) + q : ((int64_t)(slot >> 8) * ncp + q) * 256 + (slot & 255);
}

__device__ __forceinline__ void ipc_push_block(int64_t block, unsigned int nblocks, const double2 *__restrict__ f2v, const int32_t *__restrict__ send_slots,
                                               int64_t n, int ncp, const PushArgs &a, unsigned long long epoch, unsigned int *__restrict__ done) {
    const int64_t base = block * (kIpcBlock * kIpcItems) + threadIdx.x;
    int32_t slot[kIpcItems];
    double2 m[kIpcItems];
    int q[kIpcItems];                                        // unit -> (message, pair): 32-bit division, none at all for scalar messages
#pragma unroll
    for (int k = 0; k < kIpcItems; k++) {
        const int64_t i = base + k * kIpcBlock;
        const unsigned int an = (unsigned int)(ncp < 0 ? -ncp : ncp);
        const unsigned int u = (unsigned int)i, mi = an == 1 ? u : u / an;
        q[k] = an == 1 ? 0 : (int)(u - mi * an);
        slot[k] = i < n ? send_slots[mi] : -1;
    }
#pragma unroll
    for (int k = 0; k < kIpcItems; k++) if (slot[k] >= 0) m[k] = f2v[ipc_unit(slot[k], q[k], ncp)];
    __builtin_amdgcn_s_waitcnt(0);                           // the messages are in registers: no wait is left to be placed between the stores
    for (int p = 0; p < a.n_peers; p++) {                    // a message may go to several neighbours
        const int64_t off = a.send_off[p], cnt = a.send_count[p];
        double2 *dst = a.remote_area[p];
#pragma unroll
        for (int k = 0; k < kIpcItems; k++) {                // the stores of a neighbour go out back to back
            const int64_t j = base + k * kIpcBlock - off;
            if (slot[k] >= 0 && j >= 0 && j < cnt) store_sys(dst + j, m[k]);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);                           // vmcnt(0) expcnt(0) lgkmcnt(0): this wave's stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int before = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before == nblocks - 1) {                         // every workgroup's stores are complete: raise the flags
            __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int p = 0; p < a.n_peers; p++) __hip_atomic_store(a.remote_flag[p], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__device__ __forceinline__ void ipc_unpack_block(int64_t block, double2 *__restrict__ f2v, const int32_t *__restrict__ recv_slots,
                                                 const double2 *__restrict__ area, int64_t n, int ncp, const WaitArgs &w, unsigned long long epoch,
                                                 unsigned long long limit_ticks, int *__restrict__ err) {
    __shared__ int ok;
    const int64_t base = block * (kIpcBlock * kIpcItems) + threadIdx.x;
    int32_t slot[kIpcItems];
    int q[kIpcItems];
#pragma unroll
    for (int k = 0; k < kIpcItems; k++) {                    // while thread 0 waits
        const int64_t i = base + k * kIpcBlock;
        const unsigned int an = (unsigned int)(ncp < 0 ? -ncp : ncp);
        const unsigned int u = (unsigned int)i, mi = an == 1 ? u : u / an;
        q[k] = an == 1 ? 0 : (int)(u - mi * an);
        slot[k] = i < n ? recv_slots[mi] : -1;
    }
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        // a neighbour that did not arrive once is not waited for again: every later exchange would spin for the full limit (the host
        // reads the error word only at the end of a region) — the grid drains at once and the results stay void
        int good = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 ? 1 : 0;
        for (int p = 0; p < w.n_peers && good; p++)
            while (__hip_atomic_load(w.flag[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
                if (wall_clock64() - t0 > limit_ticks) { good = 0; break; }
                __builtin_amdgcn_s_sleep(8);
            }
        if (!good) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = good;
    }
    __syncthreads();                                         // the loads below are issued after the flag has been seen
    if (!ok) return;
    static_assert(kIpcItems == 4, "load_sys4");
    if (n == 0) return;
    u32x4 r[kIpcItems];
    const double2 *src[kIpcItems];
#pragma unroll
    for (int k = 0; k < kIpcItems; k++) src[k] = area + std::min<int64_t>(base + k * kIpcBlock, n - 1);   // every lane loads; the tail re-reads the last message
    load_sys4(src[0], src[1], src[2], src[3], r);
#pragma unroll
    for (int k = 0; k < kIpcItems; k++)
        if (slot[k] >= 0) { double2 v; __builtin_memcpy(&v, &r[k], 16); f2v[ipc_unit(slot[k], q[k], ncp)] = v; }
}

__global__ __launch_bounds__(kIpcBlock) void k_ipc_push(const double2 *__restrict__ f2v, const int32_t *__restrict__ send_slots, int64_t n, int ncp,
                                                        PushArgs a, unsigned long long epoch, unsigned int *__restrict__ done) {
    ipc_push_block(blockIdx.x, gridDim.x, f2v, send_slots, n, ncp, a, epoch, done);
}

__global__ __launch_bounds__(kIpcBlock) void k_ipc_unpack(double2 *__restrict__ f2v, const int32_t *__restrict__ recv_slots, const double2 *__restrict__ area,
                                                          int64_t n, int ncp, WaitArgs w, unsigned long long epoch, unsigned long long limit_ticks,
                                                          int *__restrict__ err) {
    ipc_unpack_block(blockIdx.x, f2v, recv_slots, area, n, ncp, w, epoch, limit_ticks, err);
}

// push and unpack of one exchange in ONE launch: they touch disjoint messages (what a rank sends belongs to owned variables, what it
// receives to redundant ones).  The first n_push workgroups push and never wait, the others wait for the neighbours' flags and unpack.
// REQUIREMENT: no waiting workgroup may keep a workgroup it waits for off the device.  Neither dispatch order nor residency is
// promised by the runtime, so this form is used only when (a) the caller has said that every neighbour pushes from ANOTHER device
// (cx_halo_ipc_set_fused) — a rank that is its own neighbour, handles of one process and processes that share a GPU take the two
// launches, which have no dependency inside a kernel — and (b) the whole grid fits the device at one workgroup per compute unit.
__global__ __launch_bounds__(kIpcBlock) void k_ipc_exchange(double2 *__restrict__ f2v, const int32_t *__restrict__ send_slots, int64_t n_send, int ncp,
                                                            PushArgs a, unsigned int n_push, unsigned int *__restrict__ done,
                                                            const int32_t *__restrict__ recv_slots, const double2 *__restrict__ area, int64_t n_recv,
                                                            WaitArgs w, unsigned long long epoch, unsigned long long limit_ticks, int *__restrict__ err) {
    if (blockIdx.x < n_push) ipc_push_block(blockIdx.x, n_push, f2v, send_slots, n_send, ncp, a, epoch, done);
    else ipc_unpack_block(blockIdx.x - n_push, f2v, recv_slots, area, n_recv, ncp, w, epoch, limit_ticks, err);
}

}  // namespace

namespace cx {

// 16-byte units per message (scalar: 1; dim 2 - 4: the pairs of the storage form) and the buffer they live in
static inline int ipc_ncp(const cx_handle *h) { return h->cfg.dim == 1 ? 1 : (int)(h->ncs / 2); }
// what the kernels are told: negative for the message-major records of the matrix-core dims (ipc_unit)
static inline int ipc_ncp_signed(const cx_handle *h) { return is_mfma_dim(h->cfg.dim) ? -ipc_ncp(h) : ipc_ncp(h); }
static inline double2 *ipc_messages(cx_handle *h) { return h->cfg.dim == 1 ? (double2 *)h->d_f2v : (double2 *)h->d_mv_f2v; }

void ipc_destroy(cx_handle *h) {
    for (auto &c : h->ipc_conn)
        if (c.opened) (void)hipIpcCloseMemHandle(c.opened);
    h->ipc_conn.clear();
    if (h->d_ipc_block) (void)hipFree(h->d_ipc_block);
    if (h->d_ipc_local) (void)hipFree(h->d_ipc_local);
    h->d_ipc_block = nullptr; h->d_ipc_local = nullptr; h->ipc_area_bytes = 0; h->ipc_epoch = h->ipc_pushed = 0; h->ipc_quiet_lo = 1; h->ipc_quiet_hi = 0;
}

}  // namespace cx

extern "C" {

int32_t cx_halo_ipc_alloc(cx_handle *h, void *handle64, void **local_base, int64_t *area_bytes) {
    CX_REQUIRE(h, h && h->has_graph && h->halo_state, CX_ERR_STATE, "cx_halo_ipc_alloc: call cx_halo_configure_state first");
    // (round 6: the matrix-core dims too — a message record travels as nc / 2 units)
    CX_REQUIRE(h, handle64 && local_base && area_bytes, CX_ERR_INVALID_ARGUMENT, "cx_halo_ipc_alloc: null argument");
    CX_REQUIRE(h, !h->peers.empty() && (int)h->peers.size() <= kMaxPeers, CX_ERR_STATE, "cx_halo_ipc_alloc: call cx_halo_peers first (1 to 8 neighbours)");
    CX_REQUIRE(h, (int64_t)std::max(h->send_slots.size(), h->recv_slots.size()) * cx::ipc_ncp(h) < (1ll << 31), CX_ERR_UNSUPPORTED,
               "cx_halo_ipc_alloc: a halo list of 2^31 or more 16-byte units");
    CX_HIP(h, hipSetDevice(h->cfg.device));
    CX_HIP(h, hipStreamSynchronize(h->stream));
    cx::ipc_destroy(h);
    const int64_t area = (((int64_t)h->recv_slots.size() * cx::ipc_ncp(h) * 16 + 4095) / 4096 + 1) * 4096;
    const size_t total = (size_t)(kFlagBytes + 2 * area);
    // fine-grained: coherent between agents while kernels run (what a flag protocol needs); plain hipMalloc memory is only
    // coherent at kernel boundaries
    hipError_t e = hipExtMallocWithFlags(&h->d_ipc_block, total, hipDeviceMallocFinegrained);
    if (e != hipSuccess) return fail(h, e == hipErrorOutOfMemory ? CX_ERR_OUT_OF_MEMORY : CX_ERR_DEVICE,
                                     std::string("cx_halo_ipc_alloc: hipExtMallocWithFlags(fine-grained): ") + hipGetErrorString(e));
    // any failure from here on leaves NO half-built state behind (a block without connections would pass the later checks)
#define CX_IPC_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { cx::ipc_destroy(h); \
        return fail(h, e_ == hipErrorOutOfMemory ? CX_ERR_OUT_OF_MEMORY : CX_ERR_DEVICE, std::string("cx_halo_ipc_alloc: " #call ": ") + hipGetErrorString(e_)); } } while (0)
    CX_IPC_TRY(hipMemset(h->d_ipc_block, 0, total));
    CX_IPC_TRY(hipMalloc(&h->d_ipc_local, 64));               // [0] workgroups done (push), [1] error word (unpack)
    CX_IPC_TRY(hipMemset(h->d_ipc_local, 0, 64));
    hipIpcMemHandle_t hd;
    static_assert(sizeof(hd) == 64, "hipIpcMemHandle_t is 64 bytes");
    CX_IPC_TRY(hipIpcGetMemHandle(&hd, h->d_ipc_block));
#undef CX_IPC_TRY
    std::memcpy(handle64, &hd, 64);
    h->ipc_area_bytes = area;
    h->ipc_conn.assign(h->peers.size(), cx_handle::IpcConn{});
    // the last sweep of a batch can run in two parts around an early push (cx_halo_ipc_batch): which slices hold no WRITER of a message
    // of the send list (the thread of the partner slot's variable writes it), and of those the longest run inside the owned-only slices
    h->ipc_quiet_lo = 1; h->ipc_quiet_hi = 0;
    if (h->cfg.dim == 1) {
        try { cx::haloplan::quiet_run(h); }      // cx_halo_plan.h
        catch (const std::bad_alloc &) { cx::ipc_destroy(h); return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_halo_ipc_alloc: host allocation failed"); }
    }
    *local_base = h->d_ipc_block;
    *area_bytes = area;
    return CX_OK;
}

// Peer entry `peer_index` of this handle sends to the neighbour's block: `handle64` as the neighbour's cx_halo_ipc_alloc returned
// it (another process), or `same_process_base` when the neighbour is a handle of THIS process (a block cannot be opened by the
// process that exported it).  remote_entry = index of the neighbour's peer entry that receives from this rank (its flag and its
// segment: remote_recv_off messages into the neighbour's receive area), remote_area_bytes = the neighbour's area size.
int32_t cx_halo_ipc_connect(cx_handle *h, int32_t peer_index, const void *handle64, void *same_process_base, int32_t remote_entry,
                            int64_t remote_recv_off, int64_t remote_area_bytes) {
    CX_REQUIRE(h, h && h->d_ipc_block, CX_ERR_STATE, "cx_halo_ipc_connect: call cx_halo_ipc_alloc first");
    CX_REQUIRE(h, peer_index >= 0 && peer_index < (int)h->ipc_conn.size(), CX_ERR_INVALID_ARGUMENT, "cx_halo_ipc_connect: no such peer entry");
    CX_REQUIRE(h, (handle64 != nullptr) != (same_process_base != nullptr), CX_ERR_INVALID_ARGUMENT, "cx_halo_ipc_connect: exactly one of handle64 / same_process_base");
    CX_REQUIRE(h, remote_entry >= 0 && remote_entry < kMaxPeers && remote_recv_off >= 0 && remote_area_bytes > 0 && remote_area_bytes % 4096 == 0 &&
                  (remote_recv_off + h->peers[peer_index].send_count) * cx::ipc_ncp(h) * 16 <= remote_area_bytes,
               CX_ERR_INVALID_ARGUMENT, "cx_halo_ipc_connect: segment outside the neighbour's receive area");
    CX_HIP(h, hipSetDevice(h->cfg.device));
    auto &c = h->ipc_conn[peer_index];
    if (c.opened) {          // re-connecting an entry: nobody else may still point into the mapping it opened
        for (const auto &o : h->ipc_conn)
            CX_REQUIRE(h, &o == &c || o.mapped != c.opened, CX_ERR_STATE, "cx_halo_ipc_connect: another peer entry uses the mapping this one opened; call cx_halo_ipc_alloc again");
        (void)hipIpcCloseMemHandle(c.opened); c.opened = nullptr;
    }
    char *base = (char *)same_process_base;
    c.mapped = nullptr;
    if (handle64) {
        // two peer entries may lead to the same neighbour (a ring of two ranks): a block is opened once per process
        for (const auto &o : h->ipc_conn)
            if (o.mapped && std::memcmp(o.handle, handle64, 64) == 0) { c.mapped = o.mapped; break; }
        if (!c.mapped) {
            hipIpcMemHandle_t hd;
            std::memcpy(&hd, handle64, 64);
            void *p = nullptr;
            CX_HIP(h, hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess));
            c.opened = p;
            c.mapped = p;
        }
        std::memcpy(c.handle, handle64, 64);
        base = (char *)c.mapped;
    }
    c.flag = (unsigned long long *)(base + (int64_t)remote_entry * kFlagStride);
    c.area[0] = (double2 *)(base + kFlagBytes) + remote_recv_off * cx::ipc_ncp(h);
    c.area[1] = (double2 *)(base + kFlagBytes + remote_area_bytes) + remote_recv_off * cx::ipc_ncp(h);
    c.connected = true;
    return CX_OK;
}

// The two halves of an exchange.  cx_halo_ipc_push: store this rank's boundary state into the neighbours' receive areas of the next
// epoch and raise their flags.  cx_halo_ipc_unpack: wait for the flags of the epoch last pushed and scatter the receive area into
// the redundant rows.  cx_halo_ipc_exchange = both in ONE launch.  (Handles of ONE process whose streams may share a hardware queue call
// every push before any unpack, so that no waiting unpack sits in front of the push it waits for.)
static int32_t ipc_push(cx_handle *h, const char *who, const double2 *src = nullptr) {
    CX_REQUIRE(h, h && h->has_graph && h->halo_state && h->d_ipc_block, CX_ERR_STATE, std::string(who) + ": call cx_halo_ipc_alloc first");
    CX_REQUIRE(h, h->ipc_conn.size() == h->peers.size(), CX_ERR_STATE, std::string(who) + ": the peer list changed after cx_halo_ipc_alloc");
    CX_REQUIRE(h, !h->in_sweep, CX_ERR_STATE, std::string(who) + ": a cx_sweep_begin is still open");
    CX_REQUIRE(h, h->ipc_pushed == h->ipc_epoch, CX_ERR_STATE, std::string(who) + ": the previous push has not been followed by its unpack");
    PushArgs a{};
    a.n_peers = (int)h->peers.size();
    const int ncp = cx::ipc_ncp(h);
    for (int p = 0; p < a.n_peers; p++)
        CX_REQUIRE(h, h->ipc_conn[p].connected, CX_ERR_STATE, std::string(who) + ": a peer entry is not connected (cx_halo_ipc_connect)");
    const unsigned long long epoch = (unsigned long long)(h->ipc_epoch + 1);
    const int par = (int)(epoch & 1);
    for (int p = 0; p < a.n_peers; p++) {
        a.send_off[p] = h->peers[p].send_off * ncp; a.send_count[p] = h->peers[p].send_count * ncp;
        a.remote_area[p] = h->ipc_conn[p].area[par];
        a.remote_flag[p] = h->ipc_conn[p].flag;
    }
    const int64_t ns = (int64_t)h->send_slots.size() * ncp;
    // the push runs even with nothing to send: its last workgroup raises the flags the neighbours wait for
    hipLaunchKernelGGL(k_ipc_push, dim3((unsigned)std::max<int64_t>((ns + kIpcBlock * kIpcItems - 1) / (kIpcBlock * kIpcItems), 1)), dim3(kIpcBlock), 0, h->stream,
                       src ? src : (const double2 *)cx::ipc_messages(h), (const int32_t *)h->d_send_slots, ns, cx::ipc_ncp_signed(h), a, epoch, (unsigned int *)h->d_ipc_local);
    h->ipc_pushed = (int64_t)epoch;
    CX_HIP(h, hipGetLastError());
    return CX_OK;
}

static int32_t ipc_unpack(cx_handle *h, const char *who) {
    CX_REQUIRE(h, h && h->has_graph && h->halo_state && h->d_ipc_block, CX_ERR_STATE, std::string(who) + ": call cx_halo_ipc_alloc first");
    CX_REQUIRE(h, h->ipc_conn.size() == h->peers.size(), CX_ERR_STATE, std::string(who) + ": the peer list changed after cx_halo_ipc_alloc");
    CX_REQUIRE(h, !h->in_sweep, CX_ERR_STATE, std::string(who) + ": a cx_sweep_begin is still open");
    CX_REQUIRE(h, h->ipc_pushed == h->ipc_epoch + 1, CX_ERR_STATE, std::string(who) + ": no push is waiting for its unpack (cx_halo_ipc_push)");
    WaitArgs w{};
    w.n_peers = (int)h->peers.size();
    const unsigned long long epoch = (unsigned long long)h->ipc_pushed;
    const int par = (int)(epoch & 1);
    for (int p = 0; p < w.n_peers; p++) w.flag[p] = (const unsigned long long *)((char *)h->d_ipc_block + (int64_t)p * kFlagStride);
    const int ncp = cx::ipc_ncp(h);
    const int64_t nr = (int64_t)h->recv_slots.size() * ncp;
    const double2 *area = (const double2 *)((char *)h->d_ipc_block + kFlagBytes + (int64_t)par * h->ipc_area_bytes);
    const unsigned long long limit = (unsigned long long)(h->ipc_timeout_s * 1e8);       // wall_clock64: 100 MHz
    hipLaunchKernelGGL(k_ipc_unpack, dim3((unsigned)std::max<int64_t>((nr + kIpcBlock * kIpcItems - 1) / (kIpcBlock * kIpcItems), 1)), dim3(kIpcBlock), 0, h->stream,
                       cx::ipc_messages(h), (const int32_t *)h->d_recv_slots, area, nr, cx::ipc_ncp_signed(h), w, epoch, limit, (int *)h->d_ipc_local + 1);
    h->ipc_epoch = (int64_t)epoch;
    h->sweeps_since_exchange = 0;
    CX_HIP(h, hipGetLastError());
    return CX_OK;
}

int32_t cx_halo_ipc_push(cx_handle *h) { return ipc_push(h, "cx_halo_ipc_push"); }
int32_t cx_halo_ipc_unpack(cx_handle *h) { return ipc_unpack(h, "cx_halo_ipc_unpack"); }
int32_t cx_halo_ipc_exchange(cx_handle *h) {
    const char *who = "cx_halo_ipc_exchange";
    CX_REQUIRE(h, h && h->has_graph && h->halo_state && h->d_ipc_block, CX_ERR_STATE, std::string(who) + ": call cx_halo_ipc_alloc first");
    CX_REQUIRE(h, !h->in_sweep, CX_ERR_STATE, std::string(who) + ": a cx_sweep_begin is still open");
    CX_REQUIRE(h, h->ipc_conn.size() == h->peers.size(), CX_ERR_STATE, std::string(who) + ": the peer list changed after cx_halo_ipc_alloc");
    // a push made early (cx_halo_ipc_batch: during the last sweep of the batch before) already carries the state of this moment
    if (h->ipc_pushed == h->ipc_epoch + 1) return ipc_unpack(h, who);
    CX_REQUIRE(h, h->ipc_pushed == h->ipc_epoch, CX_ERR_STATE, std::string(who) + ": push / unpack out of step");
    {
        const int64_t per_ = kIpcBlock * kIpcItems, ncp_ = cx::ipc_ncp(h);
        const int64_t wgs = std::max<int64_t>(((int64_t)h->send_slots.size() * ncp_ + per_ - 1) / per_, 1) + std::max<int64_t>(((int64_t)h->recv_slots.size() * ncp_ + per_ - 1) / per_, 1);
        int ncu = 0;
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->cfg.device);
        if (!h->ipc_fused || wgs > std::max(ncu, 1)) {      // two launches: no workgroup waits for another one of its own kernel
            const int32_t rc = ipc_push(h, who);
            return rc != CX_OK ? rc : ipc_unpack(h, who);
        }
    }
    PushArgs a{};
    WaitArgs w{};
    a.n_peers = w.n_peers = (int)h->peers.size();
    const int ncp = cx::ipc_ncp(h);
    for (int p = 0; p < a.n_peers; p++)
        CX_REQUIRE(h, h->ipc_conn[p].connected, CX_ERR_STATE, std::string(who) + ": a peer entry is not connected (cx_halo_ipc_connect)");
    const unsigned long long epoch = (unsigned long long)(h->ipc_epoch + 1);
    const int par = (int)(epoch & 1);
    for (int p = 0; p < a.n_peers; p++) {
        a.send_off[p] = h->peers[p].send_off * ncp; a.send_count[p] = h->peers[p].send_count * ncp;
        a.remote_area[p] = h->ipc_conn[p].area[par];
        a.remote_flag[p] = h->ipc_conn[p].flag;
        w.flag[p] = (const unsigned long long *)((char *)h->d_ipc_block + (int64_t)p * kFlagStride);
    }
    const int64_t ns = (int64_t)h->send_slots.size() * ncp, nr = (int64_t)h->recv_slots.size() * ncp, per = kIpcBlock * kIpcItems;
    const unsigned int n_push = (unsigned)std::max<int64_t>((ns + per - 1) / per, 1), n_unpack = (unsigned)std::max<int64_t>((nr + per - 1) / per, 1);
    const double2 *area = (const double2 *)((char *)h->d_ipc_block + kFlagBytes + (int64_t)par * h->ipc_area_bytes);
    const unsigned long long limit = (unsigned long long)(h->ipc_timeout_s * 1e8);       // wall_clock64: 100 MHz
    hipLaunchKernelGGL(k_ipc_exchange, dim3(n_push + n_unpack), dim3(kIpcBlock), 0, h->stream, cx::ipc_messages(h), (const int32_t *)h->d_send_slots, ns, cx::ipc_ncp_signed(h), a,
                       n_push, (unsigned int *)h->d_ipc_local, (const int32_t *)h->d_recv_slots, area, nr, w, epoch, limit, (int *)h->d_ipc_local + 1);
    h->ipc_epoch = h->ipc_pushed = (int64_t)epoch;
    h->sweeps_since_exchange = 0;
    CX_HIP(h, hipGetLastError());
    return CX_OK;
}

// One batch with the exchange AROUND the owned part of the first sweep, all on the handle's stream:
//     push  |  first sweep, slices of owned variables only  |  wait + unpack  |  rest of the first sweep  |  sweeps 2 .. n
// What the unpack rewrites are messages INTO redundant variables, which the slices of owned variables never read; so while this
// rank computes its interior the neighbours' pushes travel, and the wait in front of the unpack is shorter by one interior sweep.
// Bit-identical to cx_halo_ipc_exchange + cx_sweep(n) (every message is written by the same thread from the same inputs), which is
// also what runs when the sweep cannot be split (no layers set, dim > 1, another schedule).  On ONE GPU (a rank as its own neighbour)
// the flags are up before the unpack starts either way and the split only costs a launch: whether it pays between GPUs is for a
// measurement there to say (bench.py times both forms at N > 1).
int32_t cx_halo_ipc_exchange_sweep(cx_handle *h, int32_t n_sweeps) {
    CX_REQUIRE(h, h && h->has_graph && h->halo_state && h->d_ipc_block, CX_ERR_STATE, "cx_halo_ipc_exchange_sweep: call cx_halo_ipc_alloc first");
    CX_REQUIRE(h, n_sweeps >= 1, CX_ERR_INVALID_ARGUMENT, "cx_halo_ipc_exchange_sweep: n_sweeps < 1");
    const bool split = h->cfg.dim == 1 && h->cfg.schedule == CX_SCHED_FUSED && h->halo_depth > 0 && h->big_vars.empty() && h->n_kary == 0 && !h->peers.empty() &&
                       h->own_slice_hi >= h->own_slice_lo;
    if (!split) {
        const int32_t rc = cx_halo_ipc_exchange(h);
        return rc != CX_OK ? rc : cx_sweep(h, n_sweeps);
    }
    int32_t rc = CX_OK;
    if (h->ipc_pushed == h->ipc_epoch) rc = ipc_push(h, "cx_halo_ipc_exchange_sweep");      // (not when cx_halo_ipc_batch pushed it already)
    if (rc != CX_OK) return rc;
    const int L = h->halo_depth;         // sweep 1 after an exchange runs every layer <= depth
    const int lo = h->trim_hi[L] >= h->trim_lo[L] ? h->trim_lo[L] : 0, hi = h->trim_hi[L] >= h->trim_lo[L] ? h->trim_hi[L] : (int)h->nslices - 1;
    h->run_slice0 = h->own_slice_lo; h->run_nslices = h->own_slice_hi - h->own_slice_lo + 1;
    sweep_main(h, false);
    rc = ipc_unpack(h, "cx_halo_ipc_exchange_sweep");
    if (rc != CX_OK) { h->run_slice0 = 0; h->run_nslices = 0; return rc; }
    h->run_slice0 = lo; h->run_nslices = hi - lo + 1; h->run_excl_lo = h->own_slice_lo; h->run_excl_hi = h->own_slice_hi;
    sweep_main(h, false);
    h->run_excl_lo = 1; h->run_excl_hi = 0; h->run_slice0 = 0; h->run_nslices = 0;
    sweep_finish(h);
    h->sweeps_since_exchange = 1;
    CX_HIP(h, hipGetLastError());
    return n_sweeps > 1 ? cx_sweep(h, n_sweeps - 1) : CX_OK;
}

// One batch with the NEXT exchange's push inside its last sweep (the wire hidden behind compute):
//     [push, unless the batch before made it]  |  sweep 1, owned slices  |  wait + unpack  |  rest of sweep 1  |  sweeps 2 .. n - 1  |
//     last sweep: every slice that holds a writer of the send list  |  PUSH of the next exchange  |  the quiet run of that sweep
// so that between a push and the wait for it lie the quiet part of a sweep and the owned part of the next one (≈ 16 us of compute on
// a 1/8 strip of C4 where cx_halo_ipc_exchange_sweep has ≈ 9 and cx_halo_ipc_exchange none).  The push reads the buffer the running
// sweep WRITES (the Jacobi output), in which every message of the send list is final once its writers have run; a sweep computed in
// parts is bit-identical to one launch (every message is written by the same thread from the same inputs).  The state after the call:
// the next exchange is pushed but not unpacked — cx_halo_ipc_exchange / _exchange_sweep / _batch pick that up.  Falls back to
// cx_halo_ipc_exchange_sweep when the sweep cannot be split, there is no quiet run, or n_sweeps < 2.
int32_t cx_halo_ipc_batch(cx_handle *h, int32_t n_sweeps) {
    const char *who = "cx_halo_ipc_batch";
    CX_REQUIRE(h, h && h->has_graph && h->halo_state && h->d_ipc_block, CX_ERR_STATE, std::string(who) + ": call cx_halo_ipc_alloc first");
    CX_REQUIRE(h, n_sweeps >= 1, CX_ERR_INVALID_ARGUMENT, std::string(who) + ": n_sweeps < 1");
    const bool split = h->cfg.dim == 1 && h->cfg.schedule == CX_SCHED_FUSED && h->halo_depth > 0 && h->big_vars.empty() && h->n_kary == 0 && !h->peers.empty() &&
                       h->own_slice_hi >= h->own_slice_lo && h->ipc_quiet_hi >= h->ipc_quiet_lo &&
                       h->cfg.materialize_messages_to_factor == 0 && n_sweeps >= 2 && n_sweeps <= h->halo_depth;
    if (!split) return cx_halo_ipc_exchange_sweep(h, n_sweeps);
    int32_t rc;
    if (h->ipc_pushed == h->ipc_epoch) { if ((rc = ipc_push(h, who)) != CX_OK) return rc; }
    auto range_of = [&](int j, int &lo, int &hi) {      // slices sweep j after the exchange has to run (cx_sweep's trimming)
        const int L = h->halo_depth - std::min(j, h->halo_depth) + 1;
        if (h->trim_hi[L] >= h->trim_lo[L]) { lo = h->trim_lo[L]; hi = h->trim_hi[L]; } else { lo = 0; hi = (int)h->nslices - 1; }
    };
    auto reset = [&]() { h->run_excl_lo = 1; h->run_excl_hi = 0; h->run_slice0 = 0; h->run_nslices = 0; };
    // ---- sweep 1 around the unpack -----------------------------------------------------------------------------------------------
    int lo, hi;
    range_of(1, lo, hi);
    h->run_slice0 = h->own_slice_lo; h->run_nslices = h->own_slice_hi - h->own_slice_lo + 1;
    sweep_main(h, false);
    rc = ipc_unpack(h, who);
    if (rc != CX_OK) { reset(); return rc; }
    h->run_slice0 = lo; h->run_nslices = hi - lo + 1; h->run_excl_lo = h->own_slice_lo; h->run_excl_hi = h->own_slice_hi;
    sweep_main(h, false);
    reset();
    sweep_finish(h);
    h->sweeps_since_exchange = 1;
    // ---- sweeps 2 .. n - 1 ---------------------------------------------------------------------------------------------------------
    if (n_sweeps > 2 && (rc = cx_sweep(h, n_sweeps - 2)) != CX_OK) return rc;
    // ---- the last sweep around the push of the NEXT exchange ------------------------------------------------------------------------
    range_of(n_sweeps, lo, hi);
    const int qlo = std::max(h->ipc_quiet_lo, lo), qhi = std::min(h->ipc_quiet_hi, hi);
    if (qhi < qlo) {                                       // nothing quiet inside what this sweep runs
        if ((rc = cx_sweep(h, 1)) != CX_OK) return rc;
        return ipc_push(h, who);
    }
    h->run_slice0 = lo; h->run_nslices = hi - lo + 1; h->run_excl_lo = qlo; h->run_excl_hi = qhi;
    sweep_main(h, false);
    reset();
    rc = ipc_push(h, who, h->d_f2v_alt);                   // the running sweep's output buffer: sweep_finish swaps it in
    if (rc != CX_OK) return rc;
    h->run_slice0 = qlo; h->run_nslices = qhi - qlo + 1;
    sweep_main(h, false);
    reset();
    sweep_finish(h);
    h->sweeps_since_exchange++;
    CX_HIP(h, hipGetLastError());
    return CX_OK;
}

// on != 0: the caller asserts that every neighbour of this handle pushes from ANOTHER device, so that push and unpack of an exchange may
// share one launch (k_ipc_exchange's requirement); default off: two launches
int32_t cx_halo_ipc_set_fused(cx_handle *h, int32_t on) {
    CX_REQUIRE(h, h, CX_ERR_INVALID_ARGUMENT, "cx_halo_ipc_set_fused: null handle");
    h->ipc_fused = on != 0;
    return CX_OK;
}

// Waits for the stream and reports whether any unpack gave up waiting for a neighbour (then its redundant rows are stale and
// every result since is void).
int32_t cx_halo_ipc_status(cx_handle *h, int32_t *timed_out, int64_t *exchanges) {
    CX_REQUIRE(h, h && h->d_ipc_block, CX_ERR_STATE, "cx_halo_ipc_status: call cx_halo_ipc_alloc first");
    CX_REQUIRE(h, timed_out && exchanges, CX_ERR_INVALID_ARGUMENT, "cx_halo_ipc_status: null argument");
    CX_HIP(h, hipStreamSynchronize(h->stream));
    int32_t e[2] = {0, 0};
    CX_HIP(h, hipMemcpy(e, h->d_ipc_local, 8, hipMemcpyDeviceToHost));
    *timed_out = e[1];
    *exchanges = h->ipc_epoch;
    return CX_OK;
}

int32_t cx_halo_ipc_set_timeout(cx_handle *h, double seconds) {
    CX_REQUIRE(h, h, CX_ERR_INVALID_ARGUMENT, "cx_halo_ipc_set_timeout: null handle");
    CX_REQUIRE(h, seconds > 0 && seconds <= 600, CX_ERR_INVALID_ARGUMENT, "cx_halo_ipc_set_timeout: 0 < seconds <= 600");
    h->ipc_timeout_s = seconds;
    return CX_OK;
}

}  // extern "C"
