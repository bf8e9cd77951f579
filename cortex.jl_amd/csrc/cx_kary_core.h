// cx_kary_core.h — one factor→variable message of a factor with more than two edges, by ONE thread (entry = 8 * row + edge position in
// the k-ary table of cx_kary.hip): shared by k_kary_items (cx_kary.hip) and the batch kernels (cx_batch.hip), where such a message
// is an item like any other of a stage of the tree schedule — one launch per stage instead of two.
#pragma once
#include <hip/hip_runtime.h>

namespace cx {

// natural (xi, w) -> (mean, variance); point mass (y, +inf) -> (y, 0); flat (0, 0) -> (0, +inf); undefined stays NaN
__device__ __forceinline__ double2 kary_moment(double2 n) {
    if (n.y == __builtin_inf()) return make_double2(n.x, 0.0);
    if (n.y == 0.0) return make_double2(0.0, __builtin_inf());
    const double v = 1.0 / n.y;
    return make_double2(n.x * v, v);
}
__device__ __forceinline__ double2 kary_natural(double mean, double var) {
    if (var == 0.0) return make_double2(mean, __builtin_inf());
    if (var == __builtin_inf()) return make_double2(0.0, 0.0);
    const double w = 1.0 / var;
    return make_double2(mean * w, w);
}

// A value another workgroup of the SAME XCD may have stored since this kernel began (the XCD-resident cluster of cx_batch.hip:
// k_ref_cluster): read from the XCD's L2, past the compute unit's vector cache — a buffer load with the scope bit sc1 (aux = 16), 16 bytes
// at once and counted by the compiler like any load, so several are in flight together (an 8-byte atomic load at agent scope is the same
// policy at about half the rate).  COH = false: an ordinary load.  base: wave-uniform; the array must stay below 2 GiB.
typedef unsigned int cx_u4v __attribute__((ext_vector_type(4)));
typedef unsigned int cx_u2v __attribute__((ext_vector_type(2)));
typedef double cx_d2v __attribute__((ext_vector_type(2)));
template <bool COH>
__device__ __forceinline__ double2 ld2(const double2 *base, int i) {
    if (!COH) return base[i];
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)const_cast<double2 *>(base), 0, 0x7fffffff, 0x00020000);
    const cx_d2v v = __builtin_bit_cast(cx_d2v, __builtin_amdgcn_raw_buffer_load_b128(r, i * 16, 0, 16));
    return make_double2(v[0], v[1]);
}
template <bool COH>
__device__ __forceinline__ double ld1(const double *base, int i) {
    if (!COH) return base[i];
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)const_cast<double *>(base), 0, 0x7fffffff, 0x00020000);
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, i * 8, 0, 16));
}

template <bool COH = false>
__device__ __forceinline__ void kary_item(int en, const int32_t *__restrict__ kslot, const double *__restrict__ kcoef, const double *__restrict__ kqb,
                                          const double2 *__restrict__ v2f, double2 *__restrict__ f2v) {
    const int row = en >> 3, e = en & 7;
    double sm = 0.0, sv = 0.0;
    for (int r = 1; r < 8; r++) {
        const int o = 8 * row + ((e + r) & 7), s = kslot[o];
        if (s < 0) continue;
        const double2 in = kary_moment(ld2<COH>(v2f, s));
        sm += kcoef[o] * in.x;
        sv += kcoef[o] * kcoef[o] * in.y;
    }
    const double c = kcoef[en];
    const double mean = (kqb[2 * row + 1] - sm) / c, var = (kqb[2 * row] + sv) / (c * c);
    if (__builtin_isnan(mean) || __builtin_isnan(var)) return;      // a dependency is undefined: the signal is not pending
    f2v[kslot[en]] = kary_natural(mean, var);
}

}  // namespace cx
