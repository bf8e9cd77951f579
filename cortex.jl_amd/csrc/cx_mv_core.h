// cx_mv_core.h — device arithmetic of small d-dimensional Gaussian messages (d = 2, 3, 4), shared by the fused flooding
// sweep (cx_mv.hip), the chain scan (cx_mvchain.hip) and the batched per-signal kernels (cx_mvbatch.hip).
// Natural form (eta = Lambda mu, Lambda = Sigma^-1), symmetric matrices packed (upper triangle); every loop unrolled on D,
// everything in registers.  Rules and their derivation: cx_mv.hip's header.
#pragma once

#include "cx_internal.h"

namespace cx {

template <int D>
struct Msg {
    static constexpr int NT = D * (D + 1) / 2;
    static constexpr int NC = D + NT;
    double eta[D];
    double lam[NT];
};

template <int D>
__host__ __device__ constexpr int tri(int i, int j) {  // i <= j
    return i * D - i * (i - 1) / 2 + (j - i);
}

template <int D>
__device__ __forceinline__ Msg<D> msg_zero() {
    Msg<D> m;
#pragma unroll
    for (int i = 0; i < D; i++) m.eta[i] = 0.0;
#pragma unroll
    for (int i = 0; i < Msg<D>::NT; i++) m.lam[i] = 0.0;
    return m;
}

template <int D>
__device__ __forceinline__ Msg<D> msg_load(const double *__restrict__ buf, int64_t nslots, int slot) {
    Msg<D> m;
#pragma unroll
    for (int i = 0; i < D; i++) m.eta[i] = buf[(int64_t)i * nslots + slot];
#pragma unroll
    for (int i = 0; i < Msg<D>::NT; i++) m.lam[i] = buf[(int64_t)(D + i) * nslots + slot];
    return m;
}

template <int D>
__device__ __forceinline__ void msg_store(double *__restrict__ buf, int64_t nslots, int slot, const Msg<D> &m) {
#pragma unroll
    for (int i = 0; i < D; i++) buf[(int64_t)i * nslots + slot] = m.eta[i];
#pragma unroll
    for (int i = 0; i < Msg<D>::NT; i++) buf[(int64_t)(D + i) * nslots + slot] = m.lam[i];
}

// ---- message buffers (factor→variable, variable→factor): block-major pairs, 16 bytes per lane ------------------------------------
// The NC doubles of a message are stored as NCP = ceil(NC / 2) double2 "planes"; the planes of the 256 slots of one SELL block
// sit next to each other:  address(slot, plane p) = (slot / 256) * (256 * 2 * NCP) + p * 512 + (slot % 256) * 2.
// Lane <-> slot, so a wave's access to one plane is ONE contiguous kilobyte of 16-byte pieces (global_load_dwordx4), like the
// scalar path's double2 messages.  (Round 2 stored NC component planes over all slots and read them 8 bytes per lane: a pure copy
// of that shape reaches 5.3 TB/s on this part, this one 5.9: tools/lab/soa_streams.hip.)  d = 2 and 3 pad the last plane.
template <int D>
struct MsgStore {
    static constexpr int NCP = (Msg<D>::NC + 1) / 2;      // planes
    static constexpr int NCS = 2 * NCP;                   // stored doubles per slot
    static constexpr int BLOCK = kBlock * NCS;            // doubles per 256-slot block
};

template <int D>
__device__ __forceinline__ int64_t slot_offset(int slot) {
    return (int64_t)(slot >> kSliceShift) * MsgStore<D>::BLOCK + (int64_t)(slot & (kBlock - 1)) * 2;
}

// NT: the message is read once by this launch and not again before it is overwritten (the streaming reads of a sweep)
template <int D, bool NT = false>
__device__ __forceinline__ Msg<D> slot_load(const double *__restrict__ buf, int slot) {
    typedef double d2v __attribute__((ext_vector_type(2)));
    const d2v *p = reinterpret_cast<const d2v *>(buf + slot_offset<D>(slot));
    Msg<D> m;
    double c[MsgStore<D>::NCS];
#pragma unroll
    for (int q = 0; q < MsgStore<D>::NCP; q++) {
        const d2v t = NT ? __builtin_nontemporal_load(p + q * kBlock) : p[q * kBlock];
        c[2 * q] = t.x; c[2 * q + 1] = t.y;
    }
#pragma unroll
    for (int i = 0; i < D; i++) m.eta[i] = c[i];
#pragma unroll
    for (int i = 0; i < Msg<D>::NT; i++) m.lam[i] = c[D + i];
    return m;
}

template <int D>
__device__ __forceinline__ void slot_store(double *__restrict__ buf, int slot, const Msg<D> &m) {
    double2 *p = reinterpret_cast<double2 *>(buf + slot_offset<D>(slot));
    double c[MsgStore<D>::NCS];
#pragma unroll
    for (int i = 0; i < D; i++) c[i] = m.eta[i];
#pragma unroll
    for (int i = 0; i < Msg<D>::NT; i++) c[D + i] = m.lam[i];
    if (MsgStore<D>::NCS > Msg<D>::NC) c[MsgStore<D>::NCS - 1] = 0.0;
#pragma unroll
    for (int q = 0; q < MsgStore<D>::NCP; q++) p[q * kBlock] = make_double2(c[2 * q], c[2 * q + 1]);
}

// the same with nontemporal stores: marginals (written once per sweep, read by nobody on the device) use the pair form too,
// indexed by the local variable number
template <int D>
__device__ __forceinline__ void slot_store_nt(double *__restrict__ buf, int slot, const Msg<D> &m) {
    typedef double d2v __attribute__((ext_vector_type(2)));
    d2v *p = reinterpret_cast<d2v *>(buf + slot_offset<D>(slot));
    double c[MsgStore<D>::NCS];
#pragma unroll
    for (int i = 0; i < D; i++) c[i] = m.eta[i];
#pragma unroll
    for (int i = 0; i < Msg<D>::NT; i++) c[D + i] = m.lam[i];
    if (MsgStore<D>::NCS > Msg<D>::NC) c[MsgStore<D>::NCS - 1] = 0.0;
#pragma unroll
    for (int q = 0; q < MsgStore<D>::NCP; q++) { d2v t; t.x = c[2 * q]; t.y = c[2 * q + 1]; __builtin_nontemporal_store(t, p + q * kBlock); }
}

template <int D>
__device__ __forceinline__ Msg<D> msg_all_nan() {
    Msg<D> m;
#pragma unroll
    for (int i = 0; i < D; i++) m.eta[i] = __builtin_nan("");
#pragma unroll
    for (int i = 0; i < Msg<D>::NT; i++) m.lam[i] = __builtin_nan("");
    return m;
}

template <int D>
__device__ __forceinline__ void msg_add(Msg<D> &a, const Msg<D> &b) {
#pragma unroll
    for (int i = 0; i < D; i++) a.eta[i] += b.eta[i];
#pragma unroll
    for (int i = 0; i < Msg<D>::NT; i++) a.lam[i] += b.lam[i];
}

// 1/sqrt(x) to double precision: v_rsq_f64 seed + two Newton steps.  sqrt(), 1.0/x and x/y each expand to 30-40 dependent
// instructions; a d = 4 rule had 28 of them (4 sqrt, 4 reciprocals, 20 divisions in the triangular solves) — with the
// reciprocal diagonal kept from the factorisation it is 4 rsqrt and no division.
__device__ __forceinline__ double rsqrt_f64(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * __builtin_fma(-hx * y, y, 1.5);
    y = y * __builtin_fma(-hx * y, y, 1.5);
    return y;
}

// lower Cholesky factor of the packed symmetric matrix S (+ optional full symmetric P): L[i][j], j <= i
template <int D>
__device__ __forceinline__ void chol(const double (&S)[Msg<D>::NT], const double *__restrict__ P, double (&Lm)[D][D], double (&ri)[D]) {
#pragma unroll
    for (int j = 0; j < D; j++) {
        double d = S[tri<D>(j, j)] + (P ? P[j * D + j] : 0.0);
#pragma unroll
        for (int k = 0; k < j; k++) d -= Lm[j][k] * Lm[j][k];
        const double inv = rsqrt_f64(d);     // non-PD input -> NaN, which marks the message undefined
        ri[j] = inv;
        Lm[j][j] = d * inv;
#pragma unroll
        for (int i = j + 1; i < D; i++) {
            double s = S[tri<D>(j, i)] + (P ? P[i * D + j] : 0.0);
#pragma unroll
            for (int k = 0; k < j; k++) s -= Lm[i][k] * Lm[j][k];
            Lm[i][j] = s * inv;
        }
    }
}

// x <- L^-1 x   (ri = reciprocal diagonal of L)
template <int D>
__device__ __forceinline__ void fwd_solve(const double (&Lm)[D][D], const double (&ri)[D], double (&x)[D]) {
#pragma unroll
    for (int i = 0; i < D; i++) {
        double s = x[i];
#pragma unroll
        for (int k = 0; k < i; k++) s -= Lm[i][k] * x[k];
        x[i] = s * ri[i];
    }
}

// tab: P (D*D) | B (D*D) | C (D*D), row-major, full.  POINT = false: the caller knows the input is no point mass (a sum of
// factor→variable messages: the chain scan's walks)
template <int D, bool POINT = true>
__device__ __forceinline__ Msg<D> mv_rule(const Msg<D> &in, const double *__restrict__ tab) {
    const double *P = tab, *B = tab + D * D, *C = tab + 2 * D * D;
    Msg<D> out;
    if (POINT && in.lam[0] == __builtin_inf()) {  // observed datum y in eta
#pragma unroll
        for (int i = 0; i < D; i++) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < D; k++) s += B[i * D + k] * in.eta[k];
            out.eta[i] = s;
#pragma unroll
            for (int j = i; j < D; j++) out.lam[tri<D>(i, j)] = C[i * D + j];
        }
        return out;
    }
    double Lm[D][D], ri[D];
    chol<D>(in.lam, P, Lm, ri);
    double Y[D][D];  // Y[:, c] = L^-1 (row c of B)'
#pragma unroll
    for (int c = 0; c < D; c++) {
        double col[D];
#pragma unroll
        for (int k = 0; k < D; k++) col[k] = B[c * D + k];
        fwd_solve<D>(Lm, ri, col);
#pragma unroll
        for (int k = 0; k < D; k++) Y[k][c] = col[k];
    }
    double z[D];
#pragma unroll
    for (int k = 0; k < D; k++) z[k] = in.eta[k];
    fwd_solve<D>(Lm, ri, z);
#pragma unroll
    for (int i = 0; i < D; i++) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < D; k++) s += Y[k][i] * z[k];
        out.eta[i] = s;
#pragma unroll
        for (int j = i; j < D; j++) {
            double t = C[i * D + j];
#pragma unroll
            for (int k = 0; k < D; k++) t -= Y[k][i] * Y[k][j];
            out.lam[tri<D>(i, j)] = t;
        }
    }
    return out;
}

// natural -> moment (mean, packed covariance) for marginals
template <int D>
__device__ __forceinline__ Msg<D> mv_to_moment(const Msg<D> &nat) {
    double Lm[D][D], ri[D];
    chol<D>(nat.lam, nullptr, Lm, ri);
    double Li[D][D];  // columns of L^-1
#pragma unroll
    for (int c = 0; c < D; c++) {
        double e[D];
#pragma unroll
        for (int k = 0; k < D; k++) e[k] = (k == c) ? 1.0 : 0.0;
        fwd_solve<D>(Lm, ri, e);
#pragma unroll
        for (int k = 0; k < D; k++) Li[k][c] = e[k];
    }
    Msg<D> out;
#pragma unroll
    for (int i = 0; i < D; i++)
#pragma unroll
        for (int j = i; j < D; j++) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < D; k++) s += Li[k][i] * Li[k][j];
            out.lam[tri<D>(i, j)] = s;
        }
#pragma unroll
    for (int i = 0; i < D; i++) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < D; j++) s += out.lam[i <= j ? tri<D>(i, j) : tri<D>(j, i)] * nat.eta[j];
        out.eta[i] = s;
    }
    return out;
}

}  // namespace cx
