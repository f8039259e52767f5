// Birchfield-Tomasi sampling-insensitive cost, 8-tap horizontal window.
//
// Reference semantics: src/birchfield_cost.cu:26-67 (BirchfieldKernelCPU, K = 4):
//   for d in [0, min(D, x+1)):
//     cost[y][x][d] = sum_{p in [max(d, x-4), min(W, x+4))} term(p, d)      (increasing p)
//     term(p, d) = min( max(0, l - rmax, rmin - l), max(0, r - lmax, lmin - r) )
//       l = Lf[y*W + p + 1],      lmin/lmax over { l, (l + Lf[y*W+p])/2,   (l + Lf[y*W+p+2])/2 }
//       r = Rf[y*W + p - d + 1],  rmin/rmax over { r, (r + Rf[y*W+p-d])/2, (r + Rf[y*W+p-d+2])/2 }
//   where Lf/Rf are the images as flat arrays: the CPU functor uses the +1-shifted
//   indices of the CUDA functor's padded shared-memory scanline on the unpadded
//   row accessor, so taps run into the next row and, on the last
//   row, past the image, where this implementation reads 0.0f (the reference is
//   pinned with two trailing zero floats).  Entries d > x are not written (Q8).
//
// Design for gfx950: one workgroup per (row, 64-column tile).  The per-sample
// triples {value, min, max} of both scanlines are computed once and staged in LDS
// as 16-byte records; lanes span the disparity axis, so the right-image record of
// tap p for disparity d (sample p - d) is one conflict-free ds_read_b128 and every
// store of cost[y][x][d0..d0+63] is one contiguous 256-byte burst.  Each lane
// slides along the tile keeping its last 8 terms in registers and re-adds them in
// the reference's order (8 dependent adds per output: exact for any input).
// Bound: HBM stores, 4 bytes per (pixel, d).
#include "stm_common.h"

namespace stm {
namespace {

constexpr int kBtBlock = 256;
constexpr int kBtTW = 64;
constexpr int kBtK = 4;

struct BtShape {
  int H, W, D, batch;
  int ln;  // staged left records  = TW + 2K
  int rn;  // staged right records = TW + 2K + D - 1
};

__device__ __forceinline__ float flat_at(const float *__restrict__ img, long n, long idx) {
  return (idx >= 0 && idx < n) ? img[idx] : 0.0f;
}

// {centre, min, max} of the sample whose centre is flat element `centre`.
__device__ __forceinline__ float4 bt_record(const float *__restrict__ img, long n, long centre) {
  const float v = flat_at(img, n, centre);
  const float a = 0.5f * (v + flat_at(img, n, centre - 1));
  const float b = 0.5f * (v + flat_at(img, n, centre + 1));
  return make_float4(v, fminf(a, fminf(b, v)), fmaxf(a, fmaxf(b, v)), 0.0f);
}

// {centre, min, max} with the CUDA functor's indexing (src/birchfield_cost.cu:92-139): the
// sample is row[p], its neighbours row[p-1] and row[p+1], zero outside the row -- what the
// CPU functor was evidently meant to compute (STM_BT_PADDED, an opt-in extension).
__device__ __forceinline__ float4 bt_record_padded(const float *__restrict__ row, int W, int p) {
  const float v = (p >= 0 && p < W) ? row[p] : 0.0f;
  const float lo = (p - 1 >= 0 && p - 1 < W) ? row[p - 1] : 0.0f;
  const float hi = (p + 1 >= 0 && p + 1 < W) ? row[p + 1] : 0.0f;
  const float a = 0.5f * (v + lo);
  const float b = 0.5f * (v + hi);
  return make_float4(v, fminf(a, fminf(b, v)), fmaxf(a, fmaxf(b, v)), 0.0f);
}

template <bool PADDED>
__global__ __launch_bounds__(kBtBlock) void birchfield_kernel(const float *__restrict__ L_all,
                                                              const float *__restrict__ R_all,
                                                              float *__restrict__ C_all,
                                                              const BtShape S) {
  extern __shared__ __align__(16) float4 bt_smem[];
  float4 *Ls = bt_smem;         // record for p  = x0 - K + i
  float4 *Rs = bt_smem + S.ln;  // record for q  = x0 - K - (D-1) + i
  const int x0 = blockIdx.x * kBtTW;
  const int y = blockIdx.y;
  const int b = blockIdx.z;
  const long npix = (long)S.H * S.W;
  const float *L = L_all + (size_t)b * npix;
  const float *R = R_all + (size_t)b * npix;
  float *C = C_all + (size_t)b * npix * S.D;

  for (int i = threadIdx.x; i < S.ln; i += kBtBlock) {
    const int p = x0 - kBtK + i;
    if (PADDED) Ls[i] = bt_record_padded(L + (size_t)y * S.W, S.W, p);
    else Ls[i] = p >= 0 ? bt_record(L, npix, (long)y * S.W + p + 1) : make_float4(0, 0, 0, 0);
  }
  for (int i = threadIdx.x; i < S.rn; i += kBtBlock) {
    const int q = x0 - kBtK - (S.D - 1) + i;
    if (PADDED) Rs[i] = bt_record_padded(R + (size_t)y * S.W, S.W, q);
    else Rs[i] = q >= 0 ? bt_record(R, npix, (long)y * S.W + q + 1) : make_float4(0, 0, 0, 0);
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = wave_uniform(threadIdx.x >> 6);
  const int chunks = (S.D + 63) >> 6;
  const int cols = (S.W - x0) < kBtTW ? (S.W - x0) : kBtTW;
  for (int chunk = wave; chunk < chunks; chunk += kBtBlock / 64) {
    const int d = chunk * 64 + lane;
    const int rbase = (S.D - 1) - d;  // staged index of sample q = p - d is (p - x0 + K) + rbase
    // term of tap p (staged index i = p - x0 + K) for this lane's disparity; 0 when
    // the reference's loop does not visit the tap.
    auto term = [&](int i) -> float {
      const int p = x0 - kBtK + i;
      const bool in = (p >= d) && (p < S.W) && (d < S.D);
      const float4 l = Ls[i];
      const float4 r = Rs[in ? i + rbase : 0];
      const float a = fmaxf(0.0f, fmaxf(l.x - r.z, r.y - l.x));
      const float c = fmaxf(0.0f, fmaxf(r.x - l.z, l.y - r.x));
      const float t = c < a ? c : a;
      return in ? t : 0.0f;
    };
    // Ring of the last 2K terms: after step i it holds taps i-7 .. i, i.e. the
    // window [x-4, x+3] of output column x = x0 + i - 7.
    float t0 = 0.0f, t1 = term(0), t2 = term(1), t3 = term(2), t4 = term(3), t5 = term(4),
          t6 = term(5), t7 = term(6);
    for (int i = 2 * kBtK - 1; i < cols + 2 * kBtK - 1; ++i) {
      t0 = t1; t1 = t2; t2 = t3; t3 = t4; t4 = t5; t5 = t6; t6 = t7;
      t7 = term(i);
      const int x = x0 + i - (2 * kBtK - 1);
      float s = 0.0f;
      s = s + t0; s = s + t1; s = s + t2; s = s + t3;
      s = s + t4; s = s + t5; s = s + t6; s = s + t7;
      if (d <= x && d < S.D) C[((size_t)y * S.W + x) * S.D + d] = s;
    }
  }
}

}  // namespace
}  // namespace stm

extern "C" int stm_birchfield_f32(const float *left, const float *right, float *cost, int H,
                                  int W, int D, int batch, void *stream) {
  return stm_birchfield_ex_f32(left, right, cost, H, W, D, 0, batch, stream);
}

extern "C" int stm_birchfield_ex_f32(const float *left, const float *right, float *cost, int H,
                                     int W, int D, int flags, int batch, void *stream) {
  using namespace stm;
  STM_REQUIRE(left && right && cost, STM_EINVAL, "stm_birchfield_f32: null pointer");
  STM_REQUIRE(H > 0 && W > 0 && D > 0 && batch > 0, STM_EINVAL,
              "stm_birchfield_f32: non-positive size (H=%d W=%d D=%d batch=%d)", H, W, D, batch);
  STM_REQUIRE(D <= STM_MAX_DISPARITY, STM_EUNSUPPORTED, "stm_birchfield_f32: D=%d exceeds %d", D,
              STM_MAX_DISPARITY);
  // The reference iterates taps with `ushort p` (src/birchfield_cost.cu:39).
  STM_REQUIRE(W < 65535, STM_EUNSUPPORTED, "stm_birchfield_f32: W exceeds the reference's ushort tap index");
  STM_REQUIRE(H <= 65535 && batch <= 65535, STM_EUNSUPPORTED, "stm_birchfield_f32: H or batch exceeds the grid limit");
  BtShape S;
  S.H = H; S.W = W; S.D = D; S.batch = batch;
  S.ln = kBtTW + 2 * kBtK;
  S.rn = kBtTW + 2 * kBtK + D - 1;
  const size_t lds = sizeof(float4) * (size_t)(S.ln + S.rn);
  const dim3 grid((W + kBtTW - 1) / kBtTW, H, batch), block(kBtBlock);
  if (flags & STM_BT_PADDED)
    hipLaunchKernelGGL(birchfield_kernel<true>, grid, block, lds, as_stream(stream), left, right, cost, S);
  else
    hipLaunchKernelGGL(birchfield_kernel<false>, grid, block, lds, as_stream(stream), left, right, cost, S);
  return check_launch("stm_birchfield_f32");
}
