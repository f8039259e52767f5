// Shared host/device helpers for libstereomatch_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "stereomatch_hip.h"

namespace stm {

// ---- host side: error reporting ---------------------------------------------
int fail(int code, const char *fmt, ...);
int check_launch(const char *what);
inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

#define STM_REQUIRE(cond, code, ...)            \
  do {                                          \
    if (!(cond)) return stm::fail(code, __VA_ARGS__); \
  } while (0)

constexpr int kWave = 64;

#if defined(__HIPCC__)
// ---- device side: wave64 cross-lane primitives on DPP ---------------------------
// dpp_ctrl encodings (gfx9): row_shr:n = 0x110+n, wave_shl:1 = 0x130,
// wave_shr:1 = 0x138, row_mirror = 0x140, row_half_mirror = 0x141,
// row_bcast:15 = 0x142, row_bcast:31 = 0x143, quad_perm = 0x00..0xff.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_f32(float old, float src) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                         __builtin_bit_cast(int, src), CTRL,
                                         ROW_MASK, BANK_MASK, false));
}
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ int dpp_i32(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, BANK_MASK, false);
}

// Value of lane-1 (lane 0 receives `fill`).
__device__ __forceinline__ float lane_up(float v, float fill) {
  return dpp_f32<0x138>(fill, v);  // wave_shr:1
}
// Value of lane+1 (lane 63 receives `fill`).
__device__ __forceinline__ float lane_down(float v, float fill) {
  return dpp_f32<0x130>(fill, v);  // wave_shl:1
}

__device__ __forceinline__ float read_lane(float v, int lane) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// Minimum over the 64 lanes, returned wave-uniform.  NaNs are ignored by
// fminf (v_min_f32 IEEE semantics); an all-+inf wave gives +inf.
__device__ __forceinline__ float wave_min_f32(float v) {
  const float inf = __builtin_inff();
  v = fminf(v, dpp_f32<0x111>(inf, v));             // row_shr:1
  v = fminf(v, dpp_f32<0x112>(inf, v));             // row_shr:2
  v = fminf(v, dpp_f32<0x114>(inf, v));             // row_shr:4
  v = fminf(v, dpp_f32<0x118>(inf, v));             // row_shr:8  -> lane 15 of each row
  v = fminf(v, dpp_f32<0x142, 0xa>(inf, v));        // row_bcast:15 into rows 1,3
  v = fminf(v, dpp_f32<0x143, 0xc>(inf, v));        // row_bcast:31 into rows 2,3
  return read_lane(v, 63);
}

__device__ __forceinline__ int wave_min_i32(int v) {
  const int big = 0x7fffffff;
  int t;
  t = dpp_i32<0x111>(big, v); v = t < v ? t : v;
  t = dpp_i32<0x112>(big, v); v = t < v ? t : v;
  t = dpp_i32<0x114>(big, v); v = t < v ? t : v;
  t = dpp_i32<0x118>(big, v); v = t < v ? t : v;
  t = dpp_i32<0x142, 0xa>(big, v); v = t < v ? t : v;
  t = dpp_i32<0x143, 0xc>(big, v); v = t < v ? t : v;
  return __builtin_amdgcn_readlane(v, 63);
}

// Minimum within aligned groups of G lanes (G = 16, 32 or 64), result valid in
// every lane of the group.  In-row steps use DPP mirrors, cross-row steps
// ds_bpermute through __shfl_xor.
template <int G>
__device__ __forceinline__ float group_min_f32(float v) {
  v = fminf(v, dpp_f32<0xb1>(v, v));   // quad_perm [1,0,3,2]
  v = fminf(v, dpp_f32<0x4e>(v, v));   // quad_perm [2,3,0,1]
  v = fminf(v, dpp_f32<0x141>(v, v));  // row_half_mirror
  v = fminf(v, dpp_f32<0x140>(v, v));  // row_mirror
  if (G >= 32) v = fminf(v, __shfl_xor(v, 16));
  if (G >= 64) v = fminf(v, __shfl_xor(v, 32));
  return v;
}
template <int G>
__device__ __forceinline__ int group_min_i32(int v) {
  int t;
  t = dpp_i32<0xb1>(v, v); v = t < v ? t : v;
  t = dpp_i32<0x4e>(v, v); v = t < v ? t : v;
  t = dpp_i32<0x141>(v, v); v = t < v ? t : v;
  t = dpp_i32<0x140>(v, v); v = t < v ? t : v;
  if (G >= 32) { t = __shfl_xor(v, 16); v = t < v ? t : v; }
  if (G >= 64) { t = __shfl_xor(v, 32); v = t < v ? t : v; }
  return v;
}

__device__ __forceinline__ int wave_uniform(int v) {
  return __builtin_amdgcn_readfirstlane(v);
}

// ---- one pixel's disparities, VPL consecutive values per lane ---------------------
template <int VPL>
struct Row {
  float v[VPL];
};

// Loads the VPL disparities of one pixel owned by this lane; lanes/elements past
// D read as +inf so that they never win a minimum.
template <int VPL, bool VEC>
__device__ __forceinline__ Row<VPL> load_row(const float *__restrict__ base, size_t pix,
                                              int D, int d0) {
  Row<VPL> r;
  const float *p = base + pix * (size_t)D + d0;
  if constexpr (VEC) {
    if constexpr (VPL % 4 == 0) {
#pragma unroll
      for (int j = 0; j < VPL; j += 4) {
        const float4 q = *reinterpret_cast<const float4 *>(p + j);
        r.v[j] = q.x; r.v[j + 1] = q.y; r.v[j + 2] = q.z; r.v[j + 3] = q.w;
      }
    } else if constexpr (VPL == 2) {
      const float2 q = *reinterpret_cast<const float2 *>(p);
      r.v[0] = q.x; r.v[1] = q.y;
    } else {
      r.v[0] = p[0];
    }
  } else {
#pragma unroll
    for (int j = 0; j < VPL; ++j) r.v[j] = (d0 + j < D) ? p[j] : __builtin_inff();
  }
  return r;
}

template <int VPL, bool VEC>
__device__ __forceinline__ void store_row(float *__restrict__ base, size_t pix, int D, int d0,
                                          const Row<VPL> &r) {
  float *p = base + pix * (size_t)D + d0;
  if constexpr (VEC) {
    if constexpr (VPL % 4 == 0) {
#pragma unroll
      for (int j = 0; j < VPL; j += 4)
        *reinterpret_cast<float4 *>(p + j) = make_float4(r.v[j], r.v[j + 1], r.v[j + 2], r.v[j + 3]);
    } else if constexpr (VPL == 2) {
      *reinterpret_cast<float2 *>(p) = make_float2(r.v[0], r.v[1]);
    } else {
      p[0] = r.v[0];
    }
  } else {
#pragma unroll
    for (int j = 0; j < VPL; ++j)
      if (d0 + j < D) p[j] = r.v[j];
  }
}

#endif  // __HIPCC__

}  // namespace stm
