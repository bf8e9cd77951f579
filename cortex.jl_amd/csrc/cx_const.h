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

}  // namespace cx
