// SSD cost-volume fill.
//
// Reference semantics: src/ssd.cu:31-61 (SSDKernel::operator()), k = kernel_size:
//   cost[r][c][d] = sum_{kr in [max(r-k,0), min(r+k,H))} sum_{kc in [max(c-k,d), min(c+k,W))}
//                   (L[kr][kc] - R[kr][kc-d])^2        row-major, mul and add rounded separately
//   cost[r][c][d] = +inf for d > c.
//
// Design for gfx950
//   * One workgroup owns one image row r and a tile of TW output columns.  The
//     2k left scanlines [c0-k, c0+TW+k) and right scanlines [c0-k-(D-1), c0+TW+k)
//     it needs are staged in LDS once, zero-filled outside the image (a zero-filled
//     row contributes (0-0)^2 = +0, which leaves a non-negative running sum
//     unchanged, so rows need no predicate).
//   * Lanes span the disparity axis: a wave owns 64 consecutive disparities of
//     one pixel, so every store of cost[r][c][d0..d0+63] is one contiguous 256-byte
//     burst (the reference's CUDA functor writes d innermost per thread, i.e.
//     uncoalesced, include/stereomatch/kernel.hpp:24-32).  The left sample of a tap
//     is the same for all lanes: it is read once per window row by lanes 0..2k-1
//     and handed out with v_readlane; the right sample is lane-dependent
//     (kc - d) and is one conflict-free ds_read_b32.
//   * ORDERED: 4k^2 dependent mul/add per output in the reference's order --
//     bit-exact for any input; compute-bound (3 VALU per tap).
//   * BOX: vertical 2k-term sums V[kc][d], then a sliding horizontal window
//     acc += V[new] - V[old].  Re-associates the sum, so it is exact only when
//     every partial sum is an integer below 2^24; a device-side classifier
//     proves that for the actual images in AUTO mode.  ~4x fewer VALU ops and
//     store-bound (4 bytes per (pixel, d)).
#include "stm_common.h"

namespace stm {
namespace {

constexpr int kSsdBlock = 256;
constexpr int kSsdTW = 64;       // output columns per workgroup
constexpr int kSsdMaxK = 16;     // window half-size covered (2k <= 32 <= lanes used for hand-out)

struct SsdShape {
  int H, W, D, k, batch;
  int lw;  // floats per staged left row  = TW + 2k
  int rw;  // floats per staged right row = TW + 2k + D - 1
};

// ws[0] = 1 if some pixel is not an integer of magnitude < 2^24, ws[1] = min, ws[2] = max.
struct SsdClass {
  int nonint;
  int vmin;
  int vmax;
  int pad;
};

__global__ void ssd_class_init_kernel(SsdClass *ws) {
  ws->nonint = 0;
  ws->vmin = 0x7fffffff;
  ws->vmax = -0x7fffffff - 1;
  ws->pad = 0;
}

__global__ __launch_bounds__(256) void ssd_classify_kernel(const float *__restrict__ a,
                                                           const float *__restrict__ b,
                                                           size_t n, SsdClass *ws) {
  int bad = 0, lo = 0x7fffffff, hi = -0x7fffffff - 1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n;
       i += (size_t)gridDim.x * blockDim.x) {
    const float v = i < n ? a[i] : b[i - n];
    const bool ok = (fabsf(v) < 16777216.0f) && (v == truncf(v));  // NaN/inf fail the first test
    if (ok) {
      const int q = (int)v;
      lo = q < lo ? q : lo;
      hi = q > hi ? q : hi;
    } else {
      bad = 1;
    }
  }
  lo = wave_min_i32(lo);
  hi = -wave_min_i32(hi == (-0x7fffffff - 1) ? 0x7fffffff : -hi);
  const unsigned long long anybad = __ballot(bad);
  // One set of atomics per workgroup: the three words are shared by the whole grid.
  __shared__ int s_lo[4], s_hi[4], s_bad[4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_lo[wave] = lo;
    s_hi[wave] = hi;
    s_bad[wave] = anybad ? 1 : 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int blo = s_lo[0], bhi = s_hi[0], bbad = s_bad[0];
    for (int w = 1; w < 4; ++w) {
      blo = s_lo[w] < blo ? s_lo[w] : blo;
      bhi = s_hi[w] > bhi ? s_hi[w] : bhi;
      bbad |= s_bad[w];
    }
    if (bbad) atomicOr(&ws->nonint, 1);
    if (blo != 0x7fffffff) {
      atomicMin(&ws->vmin, blo);
      atomicMax(&ws->vmax, bhi);
    }
  }
}

// True when the BOX path is exact: integer-valued images and every sum of up to
// (2k)^2 squared differences stays at or below 2^24.
__device__ __forceinline__ bool ssd_box_is_exact(const SsdClass *ws, int k) {
  const int nonint = ws->nonint;
  const long long range = (long long)ws->vmax - (long long)ws->vmin;
  if (nonint || range < 0) return false;
  const long long taps = 4ll * k * k;
  // range < 2^25 here, so range^2 * taps fits comfortably in 64 bits for k <= 16.
  return range * range * taps <= 16777216ll;
}

// Stage the 2k window rows of this workgroup's tile into LDS.
__device__ __forceinline__ void ssd_stage(const float *__restrict__ L, const float *__restrict__ R,
                                          const SsdShape &S, int r, int c0, float *Ls, float *Rs) {
  const int rows = 2 * S.k;
  for (int idx = threadIdx.x; idx < rows * S.lw; idx += kSsdBlock) {
    const int j = idx / S.lw, t = idx - j * S.lw;
    const int kr = r - S.k + j, kc = c0 - S.k + t;
    Ls[idx] = (kr >= 0 && kr < S.H && kc >= 0 && kc < S.W) ? L[(size_t)kr * S.W + kc] : 0.0f;
  }
  for (int idx = threadIdx.x; idx < rows * S.rw; idx += kSsdBlock) {
    const int j = idx / S.rw, t = idx - j * S.rw;
    const int kr = r - S.k + j, x = c0 - S.k - (S.D - 1) + t;
    Rs[idx] = (kr >= 0 && kr < S.H && x >= 0 && x < S.W) ? R[(size_t)kr * S.W + x] : 0.0f;
  }
  __syncthreads();
}

// ORDERED body: every (column, 64-disparity chunk) item is one wave's job.
template <int K>  // K > 0: compile-time window half-size; K == 0: runtime S.k
__device__ __forceinline__ void ssd_ordered_body(const SsdShape &S, int r, int c0,
                                                 const float *Ls, const float *Rs,
                                                 float *__restrict__ C) {
  const int k = K > 0 ? K : S.k;
  const int lane = threadIdx.x & 63;
  const int wave = wave_uniform(threadIdx.x >> 6);
  const int chunks = (S.D + 63) >> 6;
  const int cols = (S.W - c0) < kSsdTW ? (S.W - c0) : kSsdTW;
  const float inf = __builtin_inff();
  for (int item = wave; item < cols * chunks; item += kSsdBlock / 64) {
    const int ci = item / chunks, chunk = item - ci * chunks;
    const int c = c0 + ci;
    const int d = chunk * 64 + lane;
    // Column index of tap t: kc = c - k + t; staged left column = ci + t; staged
    // right column of R[kc - d] = ci + t + (D - 1) - d.
    const int rbase = ci + (S.D - 1) - d;
    float s = 0.0f;
    // Wave-uniform test: no lane's window is clipped on the left (kc >= d for
    // all taps) or on the right (kc < W).
    const bool interior = (c - k >= chunk * 64 + 63) && (c + k <= S.W);
    if (interior) {
#pragma unroll
      for (int j = 0; j < 2 * k; ++j) {
        const float lrow = Ls[j * S.lw + ci + (lane < 2 * k ? lane : 0)];
        const float *rrow = Rs + j * S.rw + (rbase >= 0 ? rbase : 0);
#pragma unroll
        for (int t = 0; t < 2 * k; ++t) {
          const float diff = read_lane(lrow, t) - rrow[t];
          const float sq = diff * diff;
          s = s + sq;
        }
      }
    } else {
      for (int j = 0; j < 2 * k; ++j) {
        const float lrow = Ls[j * S.lw + ci + (lane < 2 * k ? lane : 0)];
        for (int t = 0; t < 2 * k; ++t) {
          const int kc = c - k + t;
          const bool in = (kc >= d) && (kc < S.W) && (d < S.D);
          const float rv = Rs[j * S.rw + (in ? rbase + t : 0)];
          const float diff = read_lane(lrow, t) - rv;
          const float sq = diff * diff;
          s = in ? s + sq : s;
        }
      }
    }
    if (d < S.D) C[((size_t)r * S.W + c) * S.D + d] = d <= c ? s : inf;
  }
}

// BOX body: a wave owns one 64-disparity chunk and slides along the tile.
__device__ __forceinline__ void ssd_box_body(const SsdShape &S, int r, int c0, const float *Ls,
                                             const float *Rs, float *ring_all,
                                             float *__restrict__ C) {
  const int k = S.k;
  const int lane = threadIdx.x & 63;
  const int wave = wave_uniform(threadIdx.x >> 6);
  const int chunks = (S.D + 63) >> 6;
  const int cols = (S.W - c0) < kSsdTW ? (S.W - c0) : kSsdTW;
  const float inf = __builtin_inff();
  float *ring = ring_all + wave * (2 * k * 64) + lane;  // ring[slot * 64]
  for (int chunk = wave; chunk < chunks; chunk += kSsdBlock / 64) {
    const int d = chunk * 64 + lane;
    const int rbase = (S.D - 1) - d;
    for (int j = 0; j < 2 * k; ++j) ring[j * 64] = 0.0f;
    float acc = 0.0f;
    int slot = 0;
    // Step t adds the column kc = c0 - k + t; after it the window [kc-2k+1, kc]
    // is complete, which is the window of output column c = kc - k + 1.
    for (int t = 0; t < cols + 2 * k - 1; ++t) {
      const int kc = c0 - k + t;
      const float lcol = Ls[(lane < 2 * k ? lane : 0) * S.lw + t];
      const bool in = (kc >= d) && (kc < S.W) && (d < S.D);
      const float *rcol = Rs + (in ? rbase + t : 0);
      float v = 0.0f;
      for (int j = 0; j < 2 * k; ++j) {
        const float diff = read_lane(lcol, j) - rcol[j * S.rw];
        v = v + diff * diff;
      }
      v = in ? v : 0.0f;
      const float old = ring[slot * 64];
      ring[slot * 64] = v;
      slot = slot + 1 == 2 * k ? 0 : slot + 1;
      acc = (acc - old) + v;
      const int c = kc - k + 1;
      if (c >= c0 && d < S.D) C[((size_t)r * S.W + c) * S.D + d] = d <= c ? acc : inf;
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(kSsdBlock) void ssd_kernel(const float *__restrict__ L_all,
                                                        const float *__restrict__ R_all,
                                                        float *__restrict__ C_all,
                                                        const SsdClass *__restrict__ ws,
                                                        const SsdShape S) {
  extern __shared__ __align__(16) float smem[];
  const int c0 = blockIdx.x * kSsdTW;
  const int r = blockIdx.y;
  const int b = blockIdx.z;
  const size_t npix = (size_t)S.H * S.W;
  const float *L = L_all + (size_t)b * npix;
  const float *R = R_all + (size_t)b * npix;
  float *C = C_all + (size_t)b * npix * S.D;
  float *Ls = smem;
  float *Rs = Ls + 2 * S.k * S.lw;
  float *ring = Rs + 2 * S.k * S.rw;
  ssd_stage(L, R, S, r, c0, Ls, Rs);
  bool box = MODE == STM_SSD_BOX;
  if (MODE == STM_SSD_AUTO) box = ssd_box_is_exact(ws, S.k);
  if (box) {
    ssd_box_body(S, r, c0, Ls, Rs, ring, C);
  } else if (S.k == 7) {
    ssd_ordered_body<7>(S, r, c0, Ls, Rs, C);
  } else {
    ssd_ordered_body<0>(S, r, c0, Ls, Rs, C);
  }
}

}  // namespace
}  // namespace stm

extern "C" size_t stm_ssd_workspace_bytes(int H, int W, int D, int kernel_size, int batch) {
  (void)H; (void)W; (void)D; (void)kernel_size; (void)batch;
  return sizeof(stm::SsdClass);
}

extern "C" int stm_ssd_f32(const float *left, const float *right, float *cost, int H, int W,
                           int D, int kernel_size, int mode, void *workspace,
                           size_t workspace_bytes, int batch, void *stream) {
  using namespace stm;
  STM_REQUIRE(left && right && cost, STM_EINVAL, "stm_ssd_f32: null pointer");
  STM_REQUIRE(H > 0 && W > 0 && D > 0 && batch > 0, STM_EINVAL,
              "stm_ssd_f32: non-positive size (H=%d W=%d D=%d batch=%d)", H, W, D, batch);
  STM_REQUIRE(kernel_size >= 1, STM_EINVAL, "stm_ssd_f32: kernel_size=%d must be >= 1", kernel_size);
  STM_REQUIRE(kernel_size <= kSsdMaxK, STM_EUNSUPPORTED,
              "stm_ssd_f32: kernel_size=%d exceeds %d", kernel_size, kSsdMaxK);
  STM_REQUIRE(D <= STM_MAX_DISPARITY, STM_EUNSUPPORTED, "stm_ssd_f32: D=%d exceeds %d", D,
              STM_MAX_DISPARITY);
  STM_REQUIRE(mode == STM_SSD_AUTO || mode == STM_SSD_ORDERED || mode == STM_SSD_BOX, STM_EINVAL,
              "stm_ssd_f32: unknown mode %d", mode);
  STM_REQUIRE(H <= 65535 && batch <= 65535, STM_EUNSUPPORTED, "stm_ssd_f32: H or batch exceeds the grid limit");
  if (mode == STM_SSD_AUTO)
    STM_REQUIRE(workspace && workspace_bytes >= sizeof(SsdClass), STM_EWORKSPACE,
                "stm_ssd_f32: AUTO mode needs a workspace of %zu bytes", sizeof(SsdClass));
  hipStream_t s = as_stream(stream);
  SsdShape S;
  S.H = H; S.W = W; S.D = D; S.k = kernel_size; S.batch = batch;
  S.lw = kSsdTW + 2 * kernel_size;
  S.rw = kSsdTW + 2 * kernel_size + D - 1;
  const size_t lds = sizeof(float) * ((size_t)2 * kernel_size * (S.lw + S.rw) +
                                      (size_t)(kSsdBlock / 64) * 2 * kernel_size * 64);
  STM_REQUIRE(lds <= 160 * 1024, STM_EUNSUPPORTED,
              "stm_ssd_f32: kernel_size=%d with D=%d needs %zu bytes of LDS", kernel_size, D, lds);
  SsdClass *ws = static_cast<SsdClass *>(workspace);
  if (mode == STM_SSD_AUTO) {
    hipLaunchKernelGGL(ssd_class_init_kernel, dim3(1), dim3(1), 0, s, ws);
    const size_t n = (size_t)batch * H * W;
    size_t blocks = (2 * n + 255) / 256;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(ssd_classify_kernel, dim3((unsigned)blocks), dim3(256), 0, s, left, right, n, ws);
  }
  const dim3 grid((W + kSsdTW - 1) / kSsdTW, H, batch), block(kSsdBlock);
  hipError_t err = hipSuccess;
  if (mode == STM_SSD_AUTO) {
    if (lds > 64 * 1024)
      err = hipFuncSetAttribute((const void *)ssd_kernel<STM_SSD_AUTO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(ssd_kernel<STM_SSD_AUTO>, grid, block, lds, s, left, right, cost, ws, S);
  } else if (mode == STM_SSD_BOX) {
    if (lds > 64 * 1024)
      err = hipFuncSetAttribute((const void *)ssd_kernel<STM_SSD_BOX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(ssd_kernel<STM_SSD_BOX>, grid, block, lds, s, left, right, cost, ws, S);
  } else {
    if (lds > 64 * 1024)
      err = hipFuncSetAttribute((const void *)ssd_kernel<STM_SSD_ORDERED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(ssd_kernel<STM_SSD_ORDERED>, grid, block, lds, s, left, right, cost, ws, S);
  }
  if (err != hipSuccess)
    return fail(STM_EHIP, "stm_ssd_f32: hipFuncSetAttribute: %s", hipGetErrorString(err));
  return check_launch("stm_ssd_f32");
}
