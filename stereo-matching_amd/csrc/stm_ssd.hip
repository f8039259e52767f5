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
  int slide_pair;  // AUTO only: 1 when an ssd_slide_kernel launch handles the exact-BOX case
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

// ORDERED body: a wave's job is a group of kOrdCB adjacent output columns x one
// 64-disparity chunk.  The columns of a group share their window samples: per window row
// the 2k + kOrdCB - 1 left samples are handed out by v_readlane from one register and the
// same number of right samples are read from LDS once, then each column runs its own
// 2k-tap sum in the reference's order (every output keeps its own dependent chain, so the
// result is the ordered sum bit for bit; only the operand fetches are shared).
constexpr int kOrdCB = 4;

template <int K>  // K > 0: compile-time window half-size; K == 0: runtime S.k
__device__ __forceinline__ void ssd_ordered_body(const SsdShape &S, int r, int c0,
                                                 const float *Ls, const float *Rs,
                                                 float *__restrict__ C) {
  const int k = K > 0 ? K : S.k;
  const int lane = threadIdx.x & 63;
  const int wave = wave_uniform(threadIdx.x >> 6);
  const int chunks = (S.D + 63) >> 6;
  const int cols = (S.W - c0) < kSsdTW ? (S.W - c0) : kSsdTW;
  const int groups = (cols + kOrdCB - 1) / kOrdCB;
  const float inf = __builtin_inff();
  for (int item = wave; item < groups * chunks; item += kSsdBlock / 64) {
    const int g = item / chunks, chunk = item - g * chunks;
    const int ci0 = g * kOrdCB;
    const int d = chunk * 64 + lane;
    const int ncol = cols - ci0 < kOrdCB ? cols - ci0 : kOrdCB;
    // Wave-uniform test: no lane's window of any column of the group is clipped on the
    // left (kc >= d for all taps) or on the right (kc < W).
    const bool interior = ncol == kOrdCB && (c0 + ci0 - k >= chunk * 64 + 63) &&
                          (c0 + ci0 + kOrdCB - 1 + k <= S.W);
    if (interior) {
      // Tap t of column q: kc = c0 + ci0 + q - k + t; staged left column ci0 + q + t, staged
      // right column of R[kc - d] = ci0 + q + t + (D - 1) - d.
      const int rbase = ci0 + (S.D - 1) - d;
      float s[kOrdCB];
#pragma unroll
      for (int q = 0; q < kOrdCB; ++q) s[q] = 0.0f;
      // One window row per trip (not unrolled: 14 rows of hoisted operand loads would
      // not fit the register file).
#pragma unroll 1
      for (int j = 0; j < 2 * k; ++j) {
        const float lrow = Ls[j * S.lw + ci0 + (lane < 2 * k + kOrdCB - 1 ? lane : 0)];
        const float *rrow = Rs + j * S.rw + (rbase >= 0 ? rbase : 0);
        if (K > 0) {
          float rv[2 * (K > 0 ? K : 1) + kOrdCB - 1];
#pragma unroll
          for (int t = 0; t < 2 * k + kOrdCB - 1; ++t) rv[t] = rrow[t];
#pragma unroll
          for (int q = 0; q < kOrdCB; ++q) {
#pragma unroll
            for (int t = 0; t < 2 * k; ++t) {
              const float diff = read_lane(lrow, q + t) - rv[q + t];
              const float sq = diff * diff;
              s[q] = s[q] + sq;
            }
          }
        } else {
          for (int q = 0; q < kOrdCB; ++q) {
            for (int t = 0; t < 2 * k; ++t) {
              const float diff = read_lane(lrow, q + t) - rrow[q + t];
              const float sq = diff * diff;
              s[q] = s[q] + sq;
            }
          }
        }
      }
      if (d < S.D) {
#pragma unroll
        for (int q = 0; q < kOrdCB; ++q)
          C[((size_t)r * S.W + c0 + ci0 + q) * S.D + d] = s[q];  // d <= c holds for interior groups
      }
    } else if (K > 0 && chunk * 64 > c0 + ci0 + kOrdCB - 1) {
      // Every disparity of the chunk exceeds every column of the group: all +inf.
      if (d < S.D)
        for (int q = 0; q < ncol; ++q) C[((size_t)r * S.W + c0 + ci0 + q) * S.D + d] = inf;
    } else if (K > 0) {
      // Clipped windows (left clip kc >= d, right image border): the same shared operand
      // fetches, plus one lane mask per window column u = q + t (tap counts iff d <= kc < W);
      // a masked-out tap adds nothing, which keeps the ordered sum intact.  Columns past the
      // tile end are computed and dropped.
      const int rbase = ci0 + (S.D - 1) - d;
      float s[kOrdCB];
#pragma unroll
      for (int q = 0; q < kOrdCB; ++q) s[q] = 0.0f;
      bool in[2 * (K > 0 ? K : 1) + kOrdCB - 1];
#pragma unroll
      for (int u = 0; u < 2 * k + kOrdCB - 1; ++u) {
        const int kc = c0 + ci0 - k + u;
        in[u] = d <= (kc < S.W ? kc : -1) && d < S.D;
      }
#pragma unroll 1
      for (int j = 0; j < 2 * k; ++j) {
        const int lcol = ci0 + (lane < 2 * k + kOrdCB - 1 ? lane : 0);
        const float lrow = Ls[j * S.lw + (lcol < S.lw ? lcol : 0)];
        const float *rrow = Rs + j * S.rw + (d < S.D ? rbase : 0);
        float rv[2 * (K > 0 ? K : 1) + kOrdCB - 1];
#pragma unroll
        for (int u = 0; u < 2 * k + kOrdCB - 1; ++u)
          rv[u] = rrow[ci0 + u < S.lw ? u : 0];
#pragma unroll
        for (int q = 0; q < kOrdCB; ++q) {
#pragma unroll
          for (int t = 0; t < 2 * k; ++t) {
            const float diff = read_lane(lrow, q + t) - rv[q + t];
            const float sq = diff * diff;
            s[q] = in[q + t] ? s[q] + sq : s[q];
          }
        }
      }
      if (d < S.D) {
#pragma unroll
        for (int q = 0; q < kOrdCB; ++q)
          if (q < ncol) C[((size_t)r * S.W + c0 + ci0 + q) * S.D + d] = d <= c0 + ci0 + q ? s[q] : inf;
      }
    } else {
      for (int q = 0; q < ncol; ++q) {
        const int ci = ci0 + q;
        const int c = c0 + ci;
        const int rbase = ci + (S.D - 1) - d;
        float s = 0.0f;
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
        if (d < S.D) C[((size_t)r * S.W + c) * S.D + d] = d <= c ? s : inf;
      }
    }
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
  bool box = MODE == STM_SSD_BOX;
  if (MODE == STM_SSD_AUTO) {
    box = ssd_box_is_exact(ws, S.k);
    // A sliding-form launch covers the exact-BOX case (k = 7, D % 64 == 0).
    if (box && S.slide_pair) return;
  }
  ssd_stage(L, R, S, r, c0, Ls, Rs);
  if (box) {
    ssd_box_body(S, r, c0, Ls, Rs, ring, C);
  } else if (S.k == 7) {
    ssd_ordered_body<7>(S, r, c0, Ls, Rs, C);
  } else {
    ssd_ordered_body<0>(S, r, c0, Ls, Rs, C);
  }
}


// ---------------------------------------------------------------------------------
// BOX, sliding form (k = 7, D a multiple of 64): the fast path for integer-valued
// images.  A wave owns a strip of kSlideCW output columns and 64 disparities and walks
// down a band of rows.  Per lane it keeps the vertical window sums
//     V[t] = sum over the 2k window rows of (L[row][kc] - R[row][kc-d])^2,  kc = c0-k+t
// for the kSlideCW + 2k - 1 columns the strip's windows touch, in registers.  Moving down
// one row adds the entering row and subtracts the leaving one (2 x 3 VALU per column);
// the outputs of a row are a sliding horizontal sum over V (2 VALU per output).  The left
// sample of a tap is wave-uniform and comes from scalar loads; the right sample
// R[row][kc-d] of lane d at column kc+1 is what lane d-1 held at column kc, so the right
// scanline lives in one register that is shifted by a DPP wave_shr:1 per column, with the
// one new element (for the chunk's first disparity) fed from a scalar load.  No LDS, no
// barriers; every store of cost[r][c][d0..d0+63] is a contiguous 256-byte burst.
// All sums are exact integers <= 2^24 under the classifier's condition, so the result is
// bit-identical to the ordered sum.  ~17 VALU per output: bound by the 4-byte store.
constexpr int kSlideK = 7;
constexpr int kSlideCW = 64;
constexpr int kSlideNV = kSlideCW + 2 * kSlideK - 1;  // 77 window columns per strip

struct SlideShape {
  int H, W, D, batch;
  int tiles, chunks, bands, band_h;
  int mode;
};

// Row samples a strip needs from one image row, in registers: the left row segment
// L[row][c0-k+16j+i] (i = lane & 15: every 16-lane row holds the same sixteen samples, handed
// out inside the subtraction by a DPP row broadcast) and the right row segment
// R[row][c0-k-(d0+63)+i] (i = lane, lane+64, lane+128) that is staged through LDS.
// Rows outside the image (or not yet part of the band's window) read as zeros on both
// sides, so they contribute (0-0)^2 = 0 and need no separate code path.
constexpr int kSlideLJ = (kSlideNV + 15) / 16;  // registers of sixteen left samples
struct SlideRow {
  float l[kSlideLJ], r0, r1, r2;
};

__device__ __forceinline__ SlideRow slide_fetch(const float *__restrict__ L,
                                                const float *__restrict__ R, int row, bool live,
                                                int W, int c0, int d0, int lane) {
  SlideRow q;
  const float *Lrow = L + (size_t)(live ? row : 0) * W;
  const float *Rrow = R + (size_t)(live ? row : 0) * W;
  const int lc = c0 - kSlideK + (lane & 15);
  const int rc = c0 - kSlideK - (d0 + 63) + lane;
#pragma unroll
  for (int j = 0; j < kSlideLJ; ++j)
    q.l[j] = (live && lc + 16 * j >= 0 && lc + 16 * j < W) ? Lrow[lc + 16 * j] : 0.0f;
  q.r0 = (live && rc >= 0 && rc < W) ? Rrow[rc] : 0.0f;
  q.r1 = (live && rc + 64 >= 0 && rc + 64 < W) ? Rrow[rc + 64] : 0.0f;
  q.r2 = (live && lane < kSlideNV + 63 - 128 && rc + 128 >= 0 && rc + 128 < W) ? Rrow[rc + 128] : 0.0f;
  return q;
}

constexpr int kSlideRW = 144;  // staged right-row floats per row (kSlideNV + 63 = 140, padded)

__device__ __forceinline__ void slide_stage(float *lds_row, const SlideRow &q, int lane) {
  lds_row[lane] = q.r0;
  lds_row[lane + 64] = q.r1;
  if (lane < kSlideRW - 128) lds_row[lane + 128] = q.r2;
}

// Eight differences L[kc] - R[kc - d] at once: the left sample of tap I0 + i is lane I0 + i of every
// 16-lane row of `l`, broadcast inside the subtraction (v_sub_f32_dpp row_newbcast) -- no
// v_readlane, no copy of a scalar into a register pair for a packed subtraction (together a
// third of the kernel's instructions).  The leading s_nop covers a VALU write of `l` just before.
#define STM_SLIDE_SUB8(I0)                                                                            \
  asm("s_nop 1\n\t"                                                                                   \
      "v_sub_f32_dpp %0, %8, %9 row_newbcast:" #I0 "+0 row_mask:0xf bank_mask:0xf\n\t"                 \
      "v_sub_f32_dpp %1, %8, %10 row_newbcast:" #I0 "+1 row_mask:0xf bank_mask:0xf\n\t"                \
      "v_sub_f32_dpp %2, %8, %11 row_newbcast:" #I0 "+2 row_mask:0xf bank_mask:0xf\n\t"                \
      "v_sub_f32_dpp %3, %8, %12 row_newbcast:" #I0 "+3 row_mask:0xf bank_mask:0xf\n\t"                \
      "v_sub_f32_dpp %4, %8, %13 row_newbcast:" #I0 "+4 row_mask:0xf bank_mask:0xf\n\t"                \
      "v_sub_f32_dpp %5, %8, %14 row_newbcast:" #I0 "+5 row_mask:0xf bank_mask:0xf\n\t"                \
      "v_sub_f32_dpp %6, %8, %15 row_newbcast:" #I0 "+6 row_mask:0xf bank_mask:0xf\n\t"                \
      "v_sub_f32_dpp %7, %8, %16 row_newbcast:" #I0 "+7 row_mask:0xf bank_mask:0xf"                     \
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]) \
      : "v"(l), "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]))
template <int I0>
__device__ __forceinline__ void slide_sub8(float (&o)[8], float l, const float (&r)[8]) {
  static_assert(I0 == 0 || I0 == 8, "a register holds sixteen samples");
  if constexpr (I0 == 0) STM_SLIDE_SUB8(0);
  else STM_SLIDE_SUB8(8);
}
#undef STM_SLIDE_SUB8

// The whole update V = (V + (le - re)^2) - (lx - rx)^2 of eight window columns (interior strips):
// written out, hipcc packs the multiplications and additions of column pairs and copies every
// operand into a register pair first -- as many v_mov as arithmetic.
#define STM_SLIDE_UPD1(N, I)                                                                  \
  "v_sub_f32_dpp %8, %10, %" #N " row_newbcast:" #I " row_mask:0xf bank_mask:0xf\n\t"         \
  "v_sub_f32_dpp %9, %11, %" #N "+8 row_newbcast:" #I " row_mask:0xf bank_mask:0xf\n\t"
template <int I0>
__device__ __forceinline__ void slide_upd8(float &v0, float &v1, float &v2, float &v3, float &v4, float &v5,
                                           float &v6, float &v7, float le, float lx, const float (&re)[8],
                                           const float (&rx)[8]) {
  static_assert(I0 == 0 || I0 == 8, "a register holds sixteen samples");
  float t0, t1;
#define STM_SLIDE_TAP(V, RE, RX, I)                                                          \
  "v_sub_f32_dpp %8, %10, " RE " row_newbcast:" I " row_mask:0xf bank_mask:0xf\n\t"           \
  "v_sub_f32_dpp %9, %11, " RX " row_newbcast:" I " row_mask:0xf bank_mask:0xf\n\t"           \
  "v_mul_f32 %8, %8, %8\n\t"                                                                 \
  "v_mul_f32 %9, %9, %9\n\t"                                                                 \
  "v_add_f32 " V ", " V ", %8\n\t"                                                           \
  "v_sub_f32 " V ", " V ", %9\n\t"
#define STM_SLIDE_BLOCK(A, B, C, D, E, F, G, H)                                               \
  asm("s_nop 1\n\t"                                                                          \
      STM_SLIDE_TAP("%0", "%12", "%20", A) STM_SLIDE_TAP("%1", "%13", "%21", B)              \
      STM_SLIDE_TAP("%2", "%14", "%22", C) STM_SLIDE_TAP("%3", "%15", "%23", D)              \
      STM_SLIDE_TAP("%4", "%16", "%24", E) STM_SLIDE_TAP("%5", "%17", "%25", F)              \
      STM_SLIDE_TAP("%6", "%18", "%26", G) STM_SLIDE_TAP("%7", "%19", "%27", H)              \
      "s_nop 0"                                                                              \
      : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "=&v"(t0), "=&v"(t1) \
      : "v"(le), "v"(lx), "v"(re[0]), "v"(re[1]), "v"(re[2]), "v"(re[3]), "v"(re[4]), "v"(re[5]), "v"(re[6]), \
        "v"(re[7]), "v"(rx[0]), "v"(rx[1]), "v"(rx[2]), "v"(rx[3]), "v"(rx[4]), "v"(rx[5]), "v"(rx[6]), "v"(rx[7]))
  if constexpr (I0 == 0) STM_SLIDE_BLOCK("0", "1", "2", "3", "4", "5", "6", "7");
  else STM_SLIDE_BLOCK("8", "9", "10", "11", "12", "13", "14", "15");
#undef STM_SLIDE_BLOCK
#undef STM_SLIDE_TAP
}
#undef STM_SLIDE_UPD1

// One row step of a strip: V += Q[entering row] - Q[leaving row].
// EDGE: the strip touches the left clip (kc >= d) or the right image border.
template <bool EDGE>
__device__ __forceinline__ void slide_step(float (&V)[kSlideNV], const SlideRow &e,
                                           const SlideRow &x, const float *lds_e,
                                           const float *lds_x, int W, int c0, int d, int lane) {
  const int kc0 = c0 - kSlideK;
#pragma unroll
  for (int t0 = 0; t0 < kSlideNV; t0 += 8) {
    float re[8], rx[8], de8[8], dx8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // (taps past the strip's last window column: any staged value, result unused)
      const int t = t0 + i < kSlideNV ? t0 + i : kSlideNV - 1;
      re[i] = lds_e[t + 63 - lane];
      rx[i] = lds_x[t + 63 - lane];
    }
    if constexpr (!EDGE) {
      float pad[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};  // (columns past the strip's last: unused)
#define STM_V(i) (t0 + (i) < kSlideNV ? V[t0 + (i) < kSlideNV ? t0 + (i) : 0] : pad[i])
      if ((t0 & 8) == 0)
        slide_upd8<0>(STM_V(0), STM_V(1), STM_V(2), STM_V(3), STM_V(4), STM_V(5), STM_V(6), STM_V(7), e.l[t0 >> 4], x.l[t0 >> 4], re, rx);
      else
        slide_upd8<8>(STM_V(0), STM_V(1), STM_V(2), STM_V(3), STM_V(4), STM_V(5), STM_V(6), STM_V(7), e.l[t0 >> 4], x.l[t0 >> 4], re, rx);
#undef STM_V
      continue;
    }
    if ((t0 & 8) == 0) {
      slide_sub8<0>(de8, e.l[t0 >> 4], re);
      slide_sub8<0>(dx8, x.l[t0 >> 4], rx);
    } else {
      slide_sub8<8>(de8, e.l[t0 >> 4], re);
      slide_sub8<8>(dx8, x.l[t0 >> 4], rx);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
    const int t = t0 + i;
    if (t >= kSlideNV) break;
    const float de = de8[i], dx = dx8[i];
    float qe = de * de;
    float qx = dx * dx;
    if (EDGE) {
      const int kc = kc0 + t;
      // Keep the limit and the compare next to their use: both are loop-invariant, and
      // hoisted out of the row loop the 77 limits and 77 lane masks would not fit the scalar
      // register file (they were spilled through v_writelane / v_readlane).
      int w = W;
      asm volatile("" : "+s"(w));
      const int lim = kc < w ? kc : -1;  // tap inside the window iff d <= kc < W
      // One select on the updated sum instead of one on each square: a column outside the
      // lane's window keeps V[t] (= 0) whatever the staged samples were.
      const float upd = (V[t] + qe) - qx;
      V[t] = d <= lim ? upd : V[t];
    } else {
      V[t] = (V[t] + qe) - qx;
    }
    }
  }
}

// NT: volumes that will not stay in the Infinity Cache anyway are stored past it.
template <bool NT>
__device__ __forceinline__ void slide_store(float *p, float v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

template <bool EDGE, bool NT>
__device__ __forceinline__ void slide_band(const float *__restrict__ L, const float *__restrict__ R,
                                           float *__restrict__ C, const SlideShape &S, int c0,
                                           int d0, int d, int lane, int r_begin, int r_end,
                                           float *lds) {
  float V[kSlideNV];
#pragma unroll
  for (int t = 0; t < kSlideNV; ++t) V[t] = 0.0f;
  const int W = S.W, H = S.H;
  const int first_row = r_begin - kSlideK;  // first row of the band's first window
  const float inf = __builtin_inff();
  float *lds_e = lds, *lds_x = lds + kSlideRW;
  // Iteration r brings V from the window of row r-1 to the window of row r:
  // rows [r-k, r+k-1] = previous + row (r+k-1) - row (r-k-1).  The first 2k-1 iterations
  // only build the first window (nothing leaves, nothing is stored).
  int r = r_begin - (2 * kSlideK - 1);
  auto live_enter = [&](int rr) { const int e = rr + kSlideK - 1; return e >= 0 && e < H && e >= first_row; };
  auto live_leave = [&](int rr) { const int x = rr - kSlideK - 1; return x >= 0 && x >= first_row; };
  SlideRow e = slide_fetch(L, R, r + kSlideK - 1, live_enter(r), W, c0, d0, lane);
  SlideRow x = slide_fetch(L, R, r - kSlideK - 1, live_leave(r), W, c0, d0, lane);
  for (; r < r_end; ++r) {
    wave_lds_handoff();  // the previous row's staged samples have been read by every lane
    slide_stage(lds_e, e, lane);
    slide_stage(lds_x, x, lane);
    wave_lds_handoff();  // lane l reads what lanes l-63..l+76 staged
    const SlideRow ce = e, cx = x;
    // Next iteration's rows are in flight while this one computes.
    e = slide_fetch(L, R, r + kSlideK, live_enter(r + 1) && r + 1 < r_end, W, c0, d0, lane);
    x = slide_fetch(L, R, r - kSlideK, live_leave(r + 1) && r + 1 < r_end, W, c0, d0, lane);
    slide_step<EDGE>(V, ce, cx, lds_e, lds_x, W, c0, d, lane);
    if (r >= r_begin) {
      // Outputs of row r: window columns [c-k, c+k) = V[c-c0 .. c-c0+2k-1].
      float acc = 0.0f;
#pragma unroll
      for (int t = 0; t < 2 * kSlideK; ++t) acc = acc + V[t];
      // Store address = wave-uniform row base (SGPRs) + a 32-bit per-lane byte offset that
      // advances by one pixel (4*D bytes) per column.  The offset is made opaque after every
      // step: the 64 column addresses are loop-invariant, and hoisted out of the row loop
      // they cost 128 SGPRs (spilled lane by lane through v_writelane / v_readlane, which
      // was most of the kernel's instruction count).
      char *out_u = reinterpret_cast<char *>(C + ((size_t)r * W + c0) * S.D + d0);
      unsigned off = 4u * (unsigned)lane;
      const unsigned pix_bytes = 4u * (unsigned)S.D;
#pragma unroll
      for (int ci = 0; ci < kSlideCW; ++ci) {
        if (ci > 0) acc = (acc - V[ci - 1]) + V[ci + 2 * kSlideK - 1];
        float *dst = reinterpret_cast<float *>(out_u + off);
        if (!EDGE) {
          slide_store<NT>(dst, acc);
        } else {
          int w = W, dd = d;
          asm volatile("" : "+s"(w));
          asm volatile("" : "+v"(dd));  // as above: nothing hoisted
          if (c0 + ci < w) slide_store<NT>(dst, dd <= c0 + ci ? acc : inf);
        }
        off += pix_bytes;
        asm volatile("" : "+v"(off));
      }
    }
  }
}

// want_box: 1 = this launch computes when BOX is exact, and exits otherwise (AUTO pairs it
// with an ORDERED launch that makes the opposite decision).
template <bool NT>
__global__ __launch_bounds__(256) void ssd_slide_kernel(const float *__restrict__ L_all,
                                                        const float *__restrict__ R_all,
                                                        float *__restrict__ C_all,
                                                        const SsdClass *__restrict__ ws,
                                                        const SlideShape S) {
  if (S.mode == STM_SSD_AUTO && !ssd_box_is_exact(ws, kSlideK)) return;
  const int lane = threadIdx.x & 63;
  int wave = wave_uniform((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int per_image = S.tiles * S.chunks * S.bands;
  if (wave >= per_image * S.batch) return;
  const int b = wave / per_image;
  wave -= b * per_image;
  const int chunk = wave % S.chunks;   // the waves of a workgroup share rows and columns
  wave /= S.chunks;
  const int tile = wave % S.tiles;
  const int band = wave / S.tiles;
  const size_t npix = (size_t)S.H * S.W;
  const float *L = L_all + (size_t)b * npix;
  const float *R = R_all + (size_t)b * npix;
  float *C = C_all + (size_t)b * npix * S.D;
  const int c0 = tile * kSlideCW;
  const int d0 = chunk * 64;
  const int d = d0 + lane;
  const int r_begin = band * S.band_h;
  const int r_end = r_begin + S.band_h < S.H ? r_begin + S.band_h : S.H;
  // Interior strip: every tap of every lane is inside its window and inside the image.
  const bool interior = (c0 - kSlideK >= d0 + 63) && (c0 + kSlideCW + kSlideK <= S.W);
  __shared__ float slide_lds[4][2 * kSlideRW];
  float *lds = slide_lds[threadIdx.x >> 6];
  if (interior)
    slide_band<false, NT>(L, R, C, S, c0, d0, d, lane, r_begin, r_end, lds);
  else
    slide_band<true, NT>(L, R, C, S, c0, d0, d, lane, r_begin, r_end, lds);
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
    const size_t cap = n > (4u << 20) ? 2048 : 512;  // one atomic triple per block
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(ssd_classify_kernel, dim3((unsigned)blocks), dim3(256), 0, s, left, right, n, ws);
  }
  // Sliding BOX form: k = 7 and whole 64-disparity chunks.
  const bool can_slide = kernel_size == kSlideK && (D % 64) == 0 && mode != STM_SSD_ORDERED;
  S.slide_pair = (can_slide && mode == STM_SSD_AUTO) ? 1 : 0;
  if (can_slide) {
    SlideShape T;
    T.H = H; T.W = W; T.D = D; T.batch = batch; T.mode = mode;
    T.tiles = (W + kSlideCW - 1) / kSlideCW;
    T.chunks = D / 64;
    // Bands short enough to fill the chip (>= ~4096 waves), tall enough to amortise the
    // 13-row warm-up of a band.
    long per_band = (long)T.tiles * T.chunks * batch;
    int band_h = 64;
    while (band_h > 16 && per_band * ((H + band_h - 1) / band_h) < 4096) band_h >>= 1;
    // A single small pair does not even fill the SIMDs at 16 rows per band (Teddy: 384 waves on
    // 1,024 SIMDs): then the kernel's time is one wave's, and shorter bands shorten it although
    // each repeats the 13-row warm-up.
    while (band_h > 4 && per_band * ((H + band_h - 1) / band_h) < 1024) band_h >>= 1;
    T.band_h = band_h;
    T.bands = (H + band_h - 1) / band_h;
    const long waves = per_band * T.bands;
    STM_REQUIRE(waves < (1l << 31), STM_EUNSUPPORTED, "stm_ssd_f32: too many strips");
    // Same rule as the aggregation: a volume that a consumer can still find in the 256 MiB
    // Infinity Cache is stored normally, a larger one with the non-temporal hint (4 % on a
    // 64-pair Teddy batch).
    const bool nt = sizeof(float) * (double)batch * H * W * D > 128.0 * 1024 * 1024;
    if (nt)
      hipLaunchKernelGGL(ssd_slide_kernel<true>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s,
                         left, right, cost, ws, T);
    else
      hipLaunchKernelGGL(ssd_slide_kernel<false>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s,
                         left, right, cost, ws, T);
    int rc = check_launch("stm_ssd_f32 (slide)");
    if (rc != STM_OK) return rc;
    if (mode == STM_SSD_BOX) return STM_OK;
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
