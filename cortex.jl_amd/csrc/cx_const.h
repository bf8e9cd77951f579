// cx_const.h — layout constants shared by the kernels and the GPU-free host logic (no HIP here: cx_flatten.h, cx_chains.h compile with gcc)
#pragma once

#include <cstdint>

namespace cx {

constexpr int kBlock = 256;        // threads per workgroup = 4 wave64 = one SELL slice of 256 variables
constexpr int kSmallDeg = 8;       // variables up to this degree live in the sliced-ELL region
constexpr int kSliceShift = 8;     // log2(kBlock)

// vinfo byte per variable
constexpr uint8_t kDegMask = 0x0f;   // degree 0..8; 15 = "big" variable (CSR region, wave-per-variable kernels)
constexpr uint8_t kBigDeg = 0x0f;
constexpr uint8_t kGhost = 0x40;     // degree-1 stand-in for a variable owned by another rank (halo import)
constexpr uint8_t kClamped = 0x80;   // observed variable: its messages are data, never recomputed

// The dimensions whose messages are message-major records eta[d] | Lambda[d][d] worked on by the f64 matrix cores, a wave per message,
// in 16 x 16 accumulator tiles (cx_mv64w.hip): 1 x 1, 2 x 2 or 4 x 4 tiles.  User dims 5 .. 63 run inside the smallest of them that
// holds them (padded with an unobserved unit random walk; cx_api.hip: cx_create) under the fused and reference-order schedules; the
// chain-scan and tree schedules have their plans written for 4 x 4 tiles and keep 64.
constexpr bool is_mfma_dim(int d) { return d == 16 || d == 32 || d == 64; }

}  // namespace cx
