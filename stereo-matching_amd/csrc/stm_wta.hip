// Winner-takes-all reduction: first minimum over the disparity axis.
// Reference semantics: src/winners_take_all.cu:21-40 (WTAKernel<kCPU>):
//   best = v[0]; idx = 0; for d in 1..D-1: if (v[d] < best) { best = v[d]; idx = d; }
// i.e. lowest index on ties, a NaN at d > 0 never wins, a NaN at d = 0 always
// wins.  (The reference's CUDA functor breaks ties the other way, :69; the
// parity target is the CPU rule.)
//
// Roofline: pure streaming read of the volume, 4 bytes per (pixel, d);
// HBM-bound.  A group of G = min(64, D/4) lanes owns one pixel and reads it
// as 16-byte vectors (one coalesced 16*G-byte burst per chunk of 4G
// disparities), so a wave covers 64/G pixels per step; the (cost, index)
// pair is reduced lexicographically first in registers, then across the group.
#include "stm_common.h"

namespace stm {
namespace {

constexpr int kWtaBlock = 256;

// Vector path: D % 4 == 0 and 16-byte aligned volume.
template <int G>
__global__ __launch_bounds__(kWtaBlock) void wta_vec_kernel(
    const float *__restrict__ vol, int32_t *__restrict__ disp, size_t npix,
    int D) {
  constexpr int kGroupsPerWave = kWave / G;
  const int lane = threadIdx.x & (kWave - 1);
  const int sub = lane / G;        // which pixel of the wave's step
  const int gl = lane % G;         // lane within the pixel's group
  const size_t wave = (size_t)blockIdx.x * (kWtaBlock / kWave) + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * (kWtaBlock / kWave);
  const float inf = __builtin_inff();
  const int nvec = D >> 2;  // float4 per pixel

  for (size_t base = wave * kGroupsPerWave; base < npix;
       base += nwaves * kGroupsPerWave) {
    const size_t pix = base + sub;
    const bool live = pix < npix;
    const float4 *row = reinterpret_cast<const float4 *>(vol + (live ? pix : 0) * (size_t)D);
    float best = inf;
    int idx = 0x7fffffff;
    float first = 0.0f;
    for (int v = gl; v < nvec; v += G) {
      float4 q = live ? row[v] : make_float4(inf, inf, inf, inf);
      const int d = v << 2;
      if (v == 0) first = q.x;
      if (idx == 0x7fffffff) idx = d;  // an all-+inf pixel must still answer 0
      if (q.x < best) { best = q.x; idx = d; }
      if (q.y < best) { best = q.y; idx = d + 1; }
      if (q.z < best) { best = q.z; idx = d + 2; }
      if (q.w < best) { best = q.w; idx = d + 3; }
    }
    const float m = group_min_f32<G>(best);
    // NaN-free equality: lanes holding the group minimum offer their index.
    int cand = (best == m) ? idx : 0x7fffffff;
    cand = group_min_i32<G>(cand);
    if (gl == 0 && live) {
      // v[0] NaN: nothing compares below it in the reference scan.
      disp[pix] = (first != first) ? 0 : cand;
    }
  }
}

// Generic path: any D, any alignment; one wave per pixel, dword loads.
__global__ __launch_bounds__(kWtaBlock) void wta_generic_kernel(
    const float *__restrict__ vol, int32_t *__restrict__ disp, size_t npix,
    int D) {
  const int lane = threadIdx.x & (kWave - 1);
  const size_t wave = (size_t)blockIdx.x * (kWtaBlock / kWave) + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * (kWtaBlock / kWave);
  const float inf = __builtin_inff();
  for (size_t pix = wave; pix < npix; pix += nwaves) {
    const float *row = vol + pix * (size_t)D;
    float best = inf;
    int idx = 0x7fffffff;
    for (int d = lane; d < D; d += kWave) {
      const float q = row[d];
      if (idx == 0x7fffffff) idx = d;
      if (q < best) { best = q; idx = d; }
    }
    const float first = row[0];
    const float m = group_min_f32<64>(best);
    int cand = (best == m) ? idx : 0x7fffffff;
    cand = group_min_i32<64>(cand);
    if (lane == 0) disp[pix] = (first != first) ? 0 : cand;
  }
}

}  // namespace
}  // namespace stm

extern "C" int stm_wta_f32(const float *volume, int32_t *disparity, int H,
                           int W, int D, int batch, void *stream) {
  using namespace stm;
  STM_REQUIRE(volume && disparity, STM_EINVAL, "stm_wta_f32: null pointer");
  STM_REQUIRE(H > 0 && W > 0 && D > 0 && batch > 0, STM_EINVAL,
              "stm_wta_f32: non-positive size (H=%d W=%d D=%d batch=%d)", H, W, D, batch);
  const size_t npix = (size_t)batch * H * W;
  hipStream_t s = as_stream(stream);
  const bool vec = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(volume) & 15) == 0);
  const int waves_per_block = kWtaBlock / kWave;
  if (vec) {
    const int G = D >= 256 ? 64 : (D >= 128 ? 32 : 16);
    const size_t groups = kWave / G;
    size_t blocks = (npix + groups * waves_per_block - 1) / (groups * waves_per_block);
    if (blocks > 256u * 16u) blocks = 256u * 16u;  // grid-stride beyond 16 blocks per CU
    if (blocks < 1) blocks = 1;
    if (G == 64)
      hipLaunchKernelGGL(wta_vec_kernel<64>, dim3((unsigned)blocks), dim3(kWtaBlock), 0, s, volume, disparity, npix, D);
    else if (G == 32)
      hipLaunchKernelGGL(wta_vec_kernel<32>, dim3((unsigned)blocks), dim3(kWtaBlock), 0, s, volume, disparity, npix, D);
    else
      hipLaunchKernelGGL(wta_vec_kernel<16>, dim3((unsigned)blocks), dim3(kWtaBlock), 0, s, volume, disparity, npix, D);
  } else {
    size_t blocks = (npix + waves_per_block - 1) / waves_per_block;
    if (blocks > 256u * 16u) blocks = 256u * 16u;
    hipLaunchKernelGGL(wta_generic_kernel, dim3((unsigned)blocks), dim3(kWtaBlock), 0, s, volume, disparity, npix, D);
  }
  return check_launch("stm_wta_f32");
}
