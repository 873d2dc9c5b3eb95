// Shared device/host helpers for libdflash_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/dflash_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define DFL_WAVE 64

// ---- error plumbing -------------------------------------------------------
void dfl_set_error(const char *fmt, ...);

#define DFL_REQUIRE(cond, ...)              \
  do {                                      \
    if (!(cond)) {                          \
      dfl_set_error(__VA_ARGS__);           \
      return DFL_EINVAL;                    \
    }                                       \
  } while (0)

#define DFL_CHECK_LAUNCH(name)                                             \
  do {                                                                     \
    hipError_t e_ = hipGetLastError();                                     \
    if (e_ != hipSuccess) {                                                \
      dfl_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return DFL_ELAUNCH;                                                  \
    }                                                                      \
  } while (0)

// ---- device helpers ---------------------------------------------------------
// Round-to-nearest-even fp32 -> bf16 via the hardware convert (keeps NaN a NaN).
__device__ __forceinline__ bf16_t f2bf(float x) { return (bf16_t)x; }
// bf16 -> fp32 through the bit pattern.  NOT `(float)b`: clang evaluates __bf16
// expressions with excess precision by default, and a float->bf16->float cast pair
// in the middle of an expression is then folded away (measured: the RoPE products
// came out unrounded).  The integer shift is opaque to that folding.
__device__ __forceinline__ float bf2f(bf16_t x) {
  return __builtin_bit_cast(float, (uint32_t)__builtin_bit_cast(unsigned short, x) << 16);
}
// value after one bf16 rounding, kept in fp32 (where torch would have stored bf16)
__device__ __forceinline__ float rbf(float x) { return bf2f((bf16_t)x); }

// 64-lane sum, result in every lane.  Row-local steps are DPP moves (VALU speed): an
// LDS-routed __shfl_xor butterfly costs ~100 cycles per step and serialised the RoPE
// prologues (6 us per 16 rows, measured with scripts/dbg_attn_stamps.py); the four row
// sums are then combined through readlane.
__device__ __forceinline__ float wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  const int iv = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
