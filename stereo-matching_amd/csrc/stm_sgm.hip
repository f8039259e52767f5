// Semiglobal aggregation, path-parallel: one wavefront walks one path.
//
// Reference semantics (CPU path, the parity target):
//   path table   src/semiglobal.cpp:18-91   (SGPixelPath::GeneratePaths)
//   recurrence   src/semiglobal.cpp:116-159 (SGMCostOperator::operator())
//   driver       src/semiglobal.cpp:191-194 (for each path: run(path); run(path.inverse()))
//   L(p0,d) = C(p0,d)
//   L(pi,d) = (C(pi,d) + min{ L(pi-1,d), L(pi-1,d-1)+P1, L(pi-1,d+1)+P1, m+P2' }) - m
//   m   = min over d in [0, D-2] of L(pi-1,d)      (BorderedBuffer::end() quirk, src/bordered_buffer.hpp:26-27)
//   P2' = max(P1, P2 / |I(pi) - I(pi-1)|)
//   out(p,d) += L(p,d) for every visit, in path-table order.
//
// Design for gfx950
//   * The disparity axis lives in the 64 lanes of a wave: lane l holds the
//     VPL = ceil(D/64) consecutive disparities l*VPL..l*VPL+VPL-1 of the previous
//     pixel in registers, so each pixel is one contiguous 4*D-byte burst (16 B
//     per lane at D = 256) and the d-1 / d+1 neighbours are a register move or
//     one DPP wave shift.  m is a DPP min-reduction with the d = D-1 element
//     masked out.
//   * The path table is never materialised: every launch covers one of the
//     eight direction families and each wave derives its start, step and length
//     from its index in closed form (the reference rebuilds and, on CUDA,
//     re-uploads the table on every call, src/semiglobal_gpu.cu:126-127).
//   * The accumulation order of the reference (->, <-, down, up, then the two
//     diagonal pairs) is kept by running the eight families as eight
//     stream-ordered launches; inside one launch every output element has
//     exactly one writer, so there are no atomics and the float sums are the
//     reference's bit for bit.  The reference's mis-stored diagonal end points
//      turn into "skip this inverse pass" / "repeat this one
//     n times" flags computed from the path index.
//   * Memory latency along the sequential walk is hidden by a register ring
//     that keeps PF pixels of cost and output rows in flight per wave.
//
// Bound: HBM.  Per pass 4N bytes of cost are read and 4N (first pass: write
// only) or 8N bytes of the running sum are moved, N = H*W*D: 92N in total
// against the algorithmic 8N (read cost once, write sum once).
#include <type_traits>

// This file is compiled twice (csrc/Makefile): as stm_sgm.o with STM_SGM_EXACT = 0, which holds
// the C entry points, and as stm_sgm_exact.o with STM_SGM_EXACT = 1, which holds the same
// kernels with the reference's treatment of NaN operands (STM_SGM_EXACT_NONFINITE, see
// sgm_step in stm_sgm_step.h) behind stm::sgm_run_exact().
#ifndef STM_SGM_EXACT
#define STM_SGM_EXACT 0
#endif

#include "stm_common.h"
#include "stm_sgm_step.h"
#include "stm_sgm_tiled.h"

namespace stm {
int sgm_run_exact(const float *cost, const float *left, float *out, int32_t *disp, int H, int W,
                  int D, float penalty1, float penalty2, int flags, void *workspace,
                  size_t workspace_bytes, int batch, void *stream, int final_mode);
namespace {


constexpr int kSgmBlock = 256;
constexpr int kPF = 4;  // pixels kept in flight per wave

enum Family : int {
  kRight = 0,      // group 1 forward   (1, 0)
  kLeft = 1,       // group 1 inverse   (-1, 0)
  kDown = 2,       // group 2 forward   (0, 1)
  kUp = 3,         // group 2 inverse   (0, -1)
  kDownRight = 4,  // groups 3+4 forward (1, 1)
  kUpLeft = 5,     // groups 3+4 inverse (-1, -1)
  kDownLeft = 6,   // groups 5+6 forward (-1, 1)
  kUpRight = 7,    // groups 5+6 inverse (1, -1)
};

struct Walk {
  int x0, y0, dx, dy, len;
  int add;    // 1: out += L along this walk; 0: the reference never runs this pass on `out`
  int spill;  // 1: L rows are also written to the workspace (re-walked diagonal, Q6)
};

// The reference's mis-stored diagonal end points make some inverse
// passes re-walk one corner diagonal instead of their own:
//   W > H : the inverses of group-3 paths i in [H, W-1] all start at (W-1, H-1) and
//           those of group-5 paths i in [H, W-1] at (0, H-1); each of the two diagonals
//           therefore receives its L (W-H+1) times, the extra W-H right after its own pass.
//   H > W+1: the inverses of group-4 paths i in [0, H-W-2] start at (W-1, H-1) and re-walk
//           the diagonal of path i = H-W-1 (H-W-1 times) BEFORE that path's forward pass.
// L is the same on every re-walk, so it is computed once, spilled to the workspace, and a
// pixel-parallel kernel (sgm_repeat_kernel) applies the repeated adds in order.
struct SgmQuirk {
  int wide_extra;  // W - H  when W > H, else 0
  int tall_extra;  // H-W-1 when H > W+1, else 0
};
__host__ __device__ inline SgmQuirk sgm_quirk(int H, int W, bool true_paths) {
  SgmQuirk q;
  q.wide_extra = (!true_paths && W > H) ? W - H : 0;
  q.tall_extra = (!true_paths && H > W + 1) ? H - W - 1 : 0;
  return q;
}

__host__ __device__ inline int sgm_family_paths(int family, int H, int W, bool true_paths) {
  switch (family) {
    case kRight: case kLeft: return H;
    case kDown: return W;
    // One extra wave walks the tall-image corner diagonal (compute + spill only) while
    // the vertical inverses run; its repeated adds land between families kUp and kDownRight.
    case kUp: return W + (sgm_quirk(H, W, true_paths).tall_extra > 0 ? 1 : 0);
    default: return W + H - 1;
  }
}



// Start, direction and length of path `w` of a family, following GeneratePaths and
// SGPixelPath::inverse() (src/semiglobal.hpp:42-45) entry by entry.
// true_paths: every inverse pass starts at the true end of its forward path (the eight
// directions as published, STM_SGM_TRUE_PATHS) instead of the reference's stored end.
__device__ __forceinline__ Walk sgm_decode(int family, int w, int H, int W, bool true_paths) {
  Walk k;
  switch (family) {
    case kRight: k = {0, w, 1, 0, W, 1, 0}; break;
    case kLeft: k = {W - 1, w, -1, 0, W, 1, 0}; break;
    case kDown: k = {w, 0, 0, 1, H, 1, 0}; break;
    case kUp:
      if (w < W) k = {w, H - 1, 0, -1, H, 1, 0};
      else k = {W - 1, H - 1, -1, -1, W, 0, 1};  // tall corner diagonal, spill only
      break;
    case kDownRight:
      if (w < W) {  // semiglobal.cpp:45-50
        k = {W - 1 - w, 0, 1, 1, imin(w, H - 1) + 1, 1, 0};
      } else {      // :52-61
        const int i = w - W;
        k = {0, i + 1, 1, 1, imin(H - 2 - i, W - 1) + 1, 1, 0};
      }
      break;
    case kUpLeft:
      if (w < W) {
        const int ey = imin(w, H - 1);
        k = {W - 1, ey, -1, -1, ey + 1, 1, 0};
        if (true_paths) k = {W - 1 - w + ey, ey, -1, -1, ey + 1, 1, 0};
        else if (w >= H) k.add = 0;                 // duplicate of w = H-1 (wide)
        else if (w == H - 1 && W > H) k.spill = 1;  // the re-walked diagonal itself
      } else {
        const int i = w - W;
        const int ex = imin(H - 2 - i, W - 1);
        k = {ex, H - 1, -1, -1, ex + 1, 1, 0};
        if (true_paths) k = {ex, i + 1 + ex, -1, -1, ex + 1, 1, 0};
        else if (W - 1 < H - 2 - i) k.add = 0;      // duplicate of i = H-W-1 (tall)
      }
      break;
    case kDownLeft:
      if (w < W) {  // :66-74
        k = {w, 0, -1, 1, imin(w + 1, H), 1, 0};
      } else {      // :76-89
        const int i = w - W;
        k = {W - 1, i + 1, -1, 1, imin(H - (i + 1), W), 1, 0};
      }
      break;
    default:  // kUpRight
      if (w < W) {
        const int len = imin(w + 1, H);
        k = {0, len - 1, 1, -1, len, 1, 0};
        if (true_paths) k = {w - len + 1, len - 1, 1, -1, len, 1, 0};
        else if (w >= H) k.add = 0;                 // stored end (0, H-1): duplicate of w = H-1
        else if (w == H - 1 && W > H) k.spill = 1;
      } else {
        const int i = w - W;
        const int len = imin(H - (i + 1), W);
        k = {W - len, i + len, 1, -1, len, 1, 0};
      }
      break;
  }
  return k;
}

// One pass along one path.  FIRST: the output row is overwritten with
// 0 + L (what accumulating into a zero-filled tensor gives) instead of
// read-modify-written.
template <int VPL, bool VEC, bool FIRST>
struct Walker {
  const float *__restrict__ cost;
  const float *__restrict__ image;
  float *__restrict__ out;
  float *__restrict__ spill;  // workspace rows of this image, or nullptr
  int D, d0;
  float p1, p2;
  StepCtx X;
  Row<VPL> prev;
  float prev_i;

  __device__ __forceinline__ void fetch(long q, Row<VPL> &c, Row<VPL> &s, float &inten) const {
    c = load_row<VPL, VEC>(cost, (size_t)q, D, d0);
    if constexpr (!FIRST) s = load_row<VPL, VEC>(out, (size_t)q, D, d0);
    inten = image[q];
  }

  __device__ __forceinline__ Row<VPL> advance(const Row<VPL> &c, float inten) {
    return sgm_step<VPL, VEC>(prev, c, sgm_penalty(inten, prev_i, p1, p2), X);
  }

  __device__ __forceinline__ void emit(long q, int i, const Row<VPL> &s, const Row<VPL> &cur,
                                       const Walk &k) const {
    if (k.add) {
      Row<VPL> o;
#pragma unroll
      for (int j = 0; j < VPL; ++j) o.v[j] = (FIRST ? 0.0f : s.v[j]) + cur.v[j];
      store_row<VPL, VEC>(out, (size_t)q, D, d0, o);
    }
    if (k.spill) store_row<VPL, VEC>(spill, (size_t)i, D, d0, cur);
  }

  __device__ __forceinline__ void run(const Walk k, int W) {
    const long step = (long)k.dy * W + k.dx;
    long pix = (long)k.y0 * W + k.x0;
    const int last = k.len - 1;

    Row<VPL> cbuf[kPF], sbuf[kPF];
    float ibuf[kPF];
    // Ring slot u holds pixel i with i % kPF == u; loads past the end of the
    // path re-read its last pixel so that the loop body is branch-free.
#pragma unroll
    for (int u = 0; u < kPF; ++u) fetch(pix + (long)imin(u, last) * step, cbuf[u], sbuf[u], ibuf[u]);

    // Pixel 0: L = C (semiglobal.cpp:123-127).
    {
      const Row<VPL> c = cbuf[0];
      const Row<VPL> s = sbuf[0];
      prev_i = ibuf[0];
      fetch(pix + (long)imin(kPF, last) * step, cbuf[0], sbuf[0], ibuf[0]);
      emit(pix, 0, s, c, k);
      prev = c;
      pix += step;
    }
    int i = 1;
    // Pixels 1 .. kPF-1 bring the ring into phase.
#pragma unroll
    for (int u = 1; u < kPF; ++u) {
      if (i < k.len) {
        const Row<VPL> c = cbuf[u];
        const Row<VPL> s = sbuf[u];
        const float inten = ibuf[u];
        fetch(pix + (long)imin(kPF, last - i) * step, cbuf[u], sbuf[u], ibuf[u]);
        const Row<VPL> cur = advance(c, inten);
        emit(pix, i, s, cur, k);
        prev = cur;
        prev_i = inten;
        pix += step;
        ++i;
      }
    }
    // Steady state: kPF pixels per trip, every slot consumed and refilled.
    for (; i + kPF <= k.len; i += kPF) {
#pragma unroll
      for (int u = 0; u < kPF; ++u) {
        const Row<VPL> c = cbuf[u];
        const Row<VPL> s = sbuf[u];
        const float inten = ibuf[u];
        fetch(pix + (long)imin(kPF, last - (i + u)) * step, cbuf[u], sbuf[u], ibuf[u]);
        const Row<VPL> cur = advance(c, inten);
        emit(pix, i + u, s, cur, k);
        prev = cur;
        prev_i = inten;
        pix += step;
      }
    }
    // Tail: fewer than kPF pixels left, already in the ring.
#pragma unroll
    for (int u = 0; u < kPF - 1; ++u) {
      if (i < k.len) {
        const Row<VPL> cur = advance(cbuf[u], ibuf[u]);
        emit(pix, i, sbuf[u], cur, k);
        prev = cur;
        prev_i = ibuf[u];
        pix += step;
        ++i;
      }
    }
  }
};

struct SgmShape {
  int H, W, D;
  float p1, p2;
  int family;
  int paths;     // waves per image in this family (sgm_family_paths)
  int batch;
  int ws_rows;   // workspace rows per image (min(H, W)), 0 when there is no workspace
  int dlim;        // disparities the per-step minimum runs over (D-1: reference, D: corrected)
  int true_paths;  // 1: inverse passes start at the true path ends
};

template <int VPL, bool VEC, bool FIRST>
__global__ __launch_bounds__(kSgmBlock) void sgm_pass_kernel(const float *__restrict__ cost_all,
                                                             const float *__restrict__ image_all,
                                                             float *__restrict__ out_all,
                                                             float *__restrict__ ws_all,
                                                             const SgmShape P) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = wave_uniform((int)(blockIdx.x * (kSgmBlock / kWave) + (threadIdx.x >> 6)));
  if (wave >= P.paths * P.batch) return;
  const int b = wave / P.paths;
  const int w = wave - b * P.paths;
  const size_t npix = (size_t)P.H * P.W;

  Walker<VPL, VEC, FIRST> wk;
  wk.cost = cost_all + (size_t)b * npix * P.D;
  wk.image = image_all + (size_t)b * npix;
  wk.out = out_all + (size_t)b * npix * P.D;
  wk.spill = ws_all ? ws_all + (size_t)b * P.ws_rows * P.D : nullptr;
  wk.D = P.D;
  wk.d0 = lane * VPL;
  wk.p1 = P.p1;
  wk.p2 = P.p2;
  wk.X.init(P.p1, lane * VPL, P.dlim, VEC && lane == kWave - 1 && P.dlim < P.D);

  Walk k = sgm_decode(P.family, w, P.H, P.W, P.true_paths != 0);
  if (!wk.spill) k.spill = 0;
  if ((k.add | k.spill) && k.len > 0) wk.run(k, P.W);
}

// Repeated adds of a spilled diagonal: out(p_i, d) = (...((out + L_i) + L_i)...) `count`
// times, one thread per (pixel of the walk, disparity); the chain per element is
// sequential, elements are independent.
struct SgmRepeat {
  int H, W, D, batch, ws_rows;
  int x0, y0, dx, dy, len, count;
};
__global__ __launch_bounds__(256) void sgm_repeat_kernel(float *__restrict__ out_all,
                                                         const float *__restrict__ ws_all,
                                                         const SgmRepeat R) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (idx >= (long)R.len * R.D) return;
  const int i = (int)(idx / R.D);
  const int d = (int)(idx - (long)i * R.D);
  const long pix = (long)(R.y0 + i * R.dy) * R.W + (R.x0 + i * R.dx);
  float *o = out_all + ((size_t)b * R.H * R.W + (size_t)pix) * R.D + d;
  const float l = ws_all[((size_t)b * R.ws_rows + i) * R.D + d];
  float acc = *o;
  for (int r = 0; r < R.count; ++r) acc = acc + l;
  *o = acc;
}

// =================================================================================
// Paired form: a direction and its inverse share one read-modify-write of the sum.
//
// The eight single-direction passes move 92N bytes because every pass re-reads and
// re-writes the running sum.  A path's forward pass F and inverse pass I touch the same
// pixels, and the reference adds them back to back (run(path); run(path.inverse())), so
//     out(p) = (out(p) + L_F(p)) + L_I(p)
// can be formed in one visit of p if L_F(p) is known while I walks back.  L_F depends on
// everything before p, so a first launch (sgm_ckpt_kernel) walks every forward path once,
// reading only the cost volume, and keeps the state after every K-th pixel (N/K floats
// per family).  The second launch per pair (sgm_pair_kernel) walks each path from its far
// end in segments of K pixels: it loads the segment's cost rows (and sum rows) into
// registers, rebuilds L_F for the segment from the checkpoint, then walks I through the
// same registers and writes the finished rows.  Same float operations in the same order as
// the single passes -> identical bits; traffic 16N for the four checkpoint walks +
// 8N + 3 x 12N for the pair launches + ~2N of checkpoints = ~62N.
// K pixels per segment (sgm_seg_len): the segment's cost rows and its sum rows (which become
// sum + L_F during the rebuild) are two register arrays of K * VPL floats per lane.
// =================================================================================
// Pixels per segment of the paired form.
__host__ __device__ constexpr int sgm_seg_len(int vpl) {
  // floats per lane one of the two segment row arrays occupies (K * VPL): 32 up to D = 128,
  // 64 above (measured with tools/ab_sgm.py: D = 128 is flat between 32 and 48 and loses at
  // 64; D = 256 gains 4-6 % from 32 to 64, another 1.5 % at 96 where D = 192 loses 2 %).
  const int floats = vpl >= 4 ? 64 : 32;
  return floats / vpl > 4 ? floats / vpl : 4;
}

struct PairShape {
  int H, W, D;
  float p1, p2;
  int batch;
  int ws_rows;            // spill rows per image (0: none)
  int pair;               // 0: (right,left) 1: (down,up) 2: (down-right,up-left) 3: (down-left,up-right)
  int paths[4];           // forward paths per image and pair
  int maxseg[4];          // checkpoint slots per path = ceil(maxlen / K) - 1
  long long ckpt_off[4];  // float offset of each pair's checkpoint block in the workspace
  int tall_extra;         // > 0: one extra wave walks the tall corner diagonal (spill only)
  int dlim;               // see SgmShape
  int true_paths;
};

__host__ __device__ inline int sgm_pair_fwd(int pair) { return pair * 2; }      // kRight, kDown, kDownRight, kDownLeft
__host__ __device__ inline int sgm_pair_inv(int pair) { return pair * 2 + 1; }  // kLeft, kUp, kUpLeft, kUpRight

// Paths per pair as indexed by the forward family (the kUp extra wave is not a pair path).
__host__ __device__ inline int sgm_pair_paths(int pair, int H, int W) {
  return pair == 0 ? H : (pair == 1 ? W : W + H - 1);
}

template <int VPL, bool VEC>
__global__ __launch_bounds__(kSgmBlock) void sgm_ckpt_kernel(const float *__restrict__ cost_all,
                                                             const float *__restrict__ image_all,
                                                             float *__restrict__ ws_all,
                                                             const PairShape P) {
  constexpr int K = sgm_seg_len(VPL);
  const int lane = threadIdx.x & (kWave - 1);
  int wave = wave_uniform((int)(blockIdx.x * (kSgmBlock / kWave) + (threadIdx.x >> 6)));
  const int per_image = P.paths[0] + P.paths[1] + P.paths[2] + P.paths[3] + (P.tall_extra > 0 ? 1 : 0);
  if (wave >= per_image * P.batch) return;
  const int b = wave / per_image;
  wave -= b * per_image;
  int pair = 0;
  while (pair < 4 && wave >= P.paths[pair]) { wave -= P.paths[pair]; ++pair; }
  const size_t npix = (size_t)P.H * P.W;
  const float *cost = cost_all + (size_t)b * npix * P.D;
  const float *image = image_all + (size_t)b * npix;
  const int D = P.D, d0 = lane * VPL;

  Walk k;
  float *ckpt = nullptr;
  float *spill = nullptr;
  if (pair < 4) {
    k = sgm_decode(sgm_pair_fwd(pair), wave, P.H, P.W, P.true_paths != 0);
    ckpt = ws_all + P.ckpt_off[pair] + ((size_t)b * P.paths[pair] + wave) * (size_t)P.maxseg[pair] * D;
  } else {
    k = sgm_decode(kUp, P.W, P.H, P.W, false);  // the tall corner diagonal, spill only
    spill = ws_all + (size_t)b * P.ws_rows * D;
  }
  if (k.len <= 0) return;
  const long step = (long)k.dy * P.W + k.dx;
  const long pix0 = (long)k.y0 * P.W + k.x0;
  const int last = k.len - 1;
  const float p1 = P.p1, p2 = P.p2;
  StepCtx X;
  X.init(P.p1, d0, P.dlim, VEC && lane == kWave - 1 && P.dlim < D);

  // P2' depends on the image only, so it is taken off the per-step dependency chain:
  // every 64 steps the lanes gather the 64 intensity pairs of the next block and do the
  // division once as a vector; a step then costs one v_readlane.  Lane l of `pen` holds
  // the penalty of step i0 + l (pixel i0+l-1 -> i0+l).
  // One gather per block (on vertical and diagonal paths every lane's sample is in another
  // cache line): the left neighbour's sample is a DPP shift, lane 0 takes the last sample of
  // the block before.
  float na, nb, pen = 0.0f;
  na = image[pix0];  // "sample -1" of the first block: pixel 0 itself
  auto pen_inputs = [&](int i0) {
    const float carry = read_lane(na, kWave - 1);
    na = image[pix0 + (long)imin(i0 + lane, last) * step];
    nb = lane_up(na, carry);
  };
  pen_inputs(0);

  // Ring slot u holds pixel i with i % kPF == u.  `pf` is the running element offset of the
  // pixel the next prefetch reads, min(i + kPF, last): two scalar adds per step instead of a
  // clamped 64-bit multiply.
  const long stepD = step * (long)D;
  long pf = pix0 * (long)D;
  Row<VPL> cbuf[kPF];
#pragma unroll
  for (int u = 0; u < kPF; ++u) {
    cbuf[u] = load_row_at<VPL, VEC>(cost, pf, D, d0);
    if (u < last) pf += stepD;
  }
  Row<VPL> prev = cbuf[0];
  if (spill) store_row<VPL, VEC>(spill, 0, D, d0, prev);
  for (int base = 0; base < k.len; base += kPF) {
    if ((base & (kWave - 1)) == 0) {  // kPF divides 64
      pen = sgm_penalty(na, nb, p1, p2);
      pen_inputs(base + kWave);
    }
#pragma unroll
    for (int u = 0; u < kPF; ++u) {
      const int i = base + u;
      if (i < k.len) {
        const Row<VPL> c = cbuf[u];
        cbuf[u] = load_row_at<VPL, VEC>(cost, pf, D, d0);
        if (i + kPF < last) pf += stepD;
        if (i > 0) {
          const float p2a = read_lane(pen, i & (kWave - 1));
          const Row<VPL> cur = sgm_step<VPL, VEC>(prev, c, p2a, X);
          prev = cur;
          if (spill) store_row<VPL, VEC>(spill, (size_t)i, D, d0, cur);
        }
        // State after pixel i = what segment (i+1)/K starts from.
        if (ckpt && ((i + 1) % K) == 0 && i + 1 < k.len)
          store_row<VPL, VEC>(ckpt, (size_t)((i + 1) / K - 1), D, d0, prev);
      }
    }
  }
}

// FINAL (last pair only): 0 = write the finished sum rows; 1 = also reduce every finished
// row to its first-minimum disparity (the winner-takes-all of src/winners_take_all.cu:21-40,
// fused so the sum is not re-read); 2 = disparity only, the finished sum is never written.
// NT: the segment's cost and sum rows are loaded, and the finished rows stored, with the
// non-temporal hint.  Every byte is touched once per launch, so on volumes larger than the
// 256 MiB Infinity Cache keeping them out of it is worth 1.5-6 % (measured, tools/ab_sgm.py);
// a single Teddy pair (cost + sum = 172 MB) is 4 % faster cached, hence the switch.  The
// checkpoint walk always loads cached: its four direction families re-read each other's
// lines (+9 % on Teddy batches with the hint).
template <int VPL, bool VEC, bool FIRST, int FINAL, bool NT>
__global__ __launch_bounds__(kSgmBlock) void sgm_pair_kernel(const float *__restrict__ cost_all,
                                                             const float *__restrict__ image_all,
                                                             float *__restrict__ out_all,
                                                             float *__restrict__ ws_all,
                                                             int32_t *__restrict__ disp_all,
                                                             const PairShape P) {
  constexpr int K = sgm_seg_len(VPL);
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = wave_uniform((int)(blockIdx.x * (kSgmBlock / kWave) + (threadIdx.x >> 6)));
  const int paths = P.paths[P.pair];
  if (wave >= paths * P.batch) return;
  const int b = wave / paths;
  const int w = wave - b * paths;
  const size_t npix = (size_t)P.H * P.W;
  const float *cost = cost_all + (size_t)b * npix * P.D;
  const float *image = image_all + (size_t)b * npix;
  float *out = out_all + (size_t)b * npix * P.D;
  const int D = P.D, d0 = lane * VPL;
  const float p1 = P.p1, p2 = P.p2;
  StepCtx X;
  X.init(P.p1, d0, P.dlim, VEC && lane == kWave - 1 && P.dlim < D);

  const Walk F = sgm_decode(sgm_pair_fwd(P.pair), w, P.H, P.W, P.true_paths != 0);
  const Walk I = sgm_decode(sgm_pair_inv(P.pair), w, P.H, P.W, P.true_paths != 0);
  if (F.len <= 0) return;
  const bool add_inv = I.add != 0;  // 0: the reference's inverse pass re-walks another diagonal
  float *spill = (I.spill && P.ws_rows > 0) ? ws_all + (size_t)b * P.ws_rows * D : nullptr;
  const float *ckpt = ws_all + P.ckpt_off[P.pair] + ((size_t)b * paths + w) * (size_t)P.maxseg[P.pair] * D;
  const long step = (long)F.dy * P.W + F.dx;
  const long pix0 = (long)F.y0 * P.W + F.x0;
  const int nseg = (F.len + K - 1) / K;

  Row<VPL> inv_prev;
#pragma unroll
  for (int j = 0; j < VPL; ++j) inv_prev.v[j] = 0.0f;

  // One segment.  FULL (n == K, every segment but possibly the path's last one): no per-pixel
  // bounds tests and no clamped row indices; row addresses are running 64-bit offsets (two
  // scalar adds per row).  The scalar unit issues at the VALU's rate, and at D <= 128 the
  // generic form spent twice as many scalar as vector instructions per step.
  const long stepD = step * (long)D;  // elements between consecutive pixels of the path
  auto segment = [&](auto full_tag, const int s) {
    constexpr bool FULL = decltype(full_tag)::value;
    const int base = s * K;
    const int n = FULL ? K : imin(K, F.len - base);
    const long off0 = (pix0 + (long)base * step) * (long)D;  // element offset of pixel `base`
    // sseg: the sum rows on arrival, sum + L_F after the forward rebuild (no third array:
    // 2 x K*VPL floats per lane instead of 3 x keeps one more wave per SIMD resident).
    Row<VPL> cseg[K], sseg[K];
    // Bulk loads of the segment: everything a wave needs for K pixels is in flight at once.
    {
      long off = off0;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        cseg[j] = load_row_at<VPL, VEC, NT>(cost, off, D, d0);
        if (FULL || j < n - 1) off += stepD;  // rows past the end re-read the last one
      }
    }
    // Penalties of the K + 1 pixel boundaries the segment touches, one division for all:
    // lane l holds P2' between pixels base+l-1 and base+l; the forward step into pixel
    // base+j reads lane j, the inverse step into it (from base+j+1) lane j+1.
    float pen;
    {
      // One gather: lane l reads pixel base+l-1 (clamped to the path), its right neighbour's
      // sample arrives by a DPP shift.
      const int l = imin(lane, K + 1);
      const int ig = imin(base + l > 0 ? base + l - 1 : 0, F.len - 1);
      const float g = image[pix0 + (long)ig * step];
      pen = sgm_penalty(lane_down(g, g), g, p1, p2);
    }
    if constexpr (!FIRST) {
      long off = off0;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        sseg[j] = load_row_at<VPL, VEC, NT>(out, off, D, d0);
        if (FULL || j < n - 1) off += stepD;
      }
    }
    // Forward values of the segment, from the path start (s = 0) or the checkpoint.
    Row<VPL> fprev;
    if (s == 0) {
      fprev = cseg[0];
    } else {
      fprev = load_row<VPL, VEC>(ckpt, (size_t)(s - 1), D, d0);
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if (FULL || j < n) {
        if (j > 0 || s != 0)
          fprev = sgm_step<VPL, VEC>(fprev, cseg[j], read_lane(pen, j), X);
#pragma unroll
        for (int e = 0; e < VPL; ++e)
          sseg[j].v[e] = (FIRST ? 0.0f : sseg[j].v[e]) + fprev.v[e];
      }
    }
    // Inverse walk through the same registers, finishing each row.
    long off = off0 + (long)(K - 1) * stepD;  // row K-1; rows j >= n are skipped below
    long q = pix0 + (long)(base + K - 1) * step;
#pragma unroll
    for (int j = K - 1; j >= 0; --j) {
      if (FULL || j < n) {
        Row<VPL> o;
#pragma unroll
        for (int e = 0; e < VPL; ++e) o.v[e] = sseg[j].v[e];
        if (add_inv) {
          Row<VPL> li;
          if (j == n - 1 && s == nseg - 1) {
            li = cseg[j];  // first pixel of the inverse pass: L = C
          } else {
            li = sgm_step<VPL, VEC>(inv_prev, cseg[j], read_lane(pen, j + 1), X);
          }
          inv_prev = li;
#pragma unroll
          for (int e = 0; e < VPL; ++e) o.v[e] = o.v[e] + li.v[e];
          if (spill) store_row<VPL, VEC>(spill, (size_t)(F.len - 1 - (base + j)), D, d0, li);
        }
        // (The re-walked diagonal keeps its rows even in disparity-only mode: the repeat
        // kernel finishes them.)
        if (FINAL != 2 || spill) store_row_at<VPL, VEC, NT>(out, off, D, d0, o);
        if constexpr (FINAL != 0) {
          // Strict-'<' scan in register order, then lowest index among the lanes that hold
          // the wave minimum; a NaN at d = 0 wins, as in the reference scan.
          float best = __builtin_inff();
          int idx = (VEC || d0 < D) ? d0 : 0x7fffffff;  // VEC: D = 64 * VPL, every lane is live
#pragma unroll
          for (int e = 0; e < VPL; ++e)
            if ((VEC || d0 + e < D) && o.v[e] < best) { best = o.v[e]; idx = d0 + e; }
          // Lanes hold increasing disparities, so the first lane whose value equals the wave
          // minimum holds the lowest winning index (best is never NaN: it starts at +inf).
          const float m = wave_min_f32(best);
          const int win = __builtin_amdgcn_readlane(idx, __builtin_ctzll(__ballot(best == m)));
          const float first = read_lane(o.v[0], 0);
          if (lane == 0) disp_all[(size_t)b * npix + (size_t)q] = (first != first) ? 0 : win;
        }
      }
      off -= stepD;
      q -= step;
    }
  };

  for (int s = nseg - 1; s >= 0; --s) {
    if (F.len - s * K >= K) segment(std::true_type{}, s);
    else segment(std::false_type{}, s);
  }
}

// The same repeated adds, one wave per pixel of the diagonal, followed by the pixel's
// winner-takes-all (fused aggregation + reduction on wide images).  store_sum = 0 leaves
// the sum volume untouched.
__global__ __launch_bounds__(64) void sgm_repeat_wta_kernel(float *__restrict__ out_all,
                                                            const float *__restrict__ ws_all,
                                                            int32_t *__restrict__ disp_all,
                                                            const SgmRepeat R, int store_sum) {
  const int lane = threadIdx.x;
  const int i = blockIdx.x;
  const int b = blockIdx.y;
  const long pix = (long)(R.y0 + i * R.dy) * R.W + (R.x0 + i * R.dx);
  float *o = out_all + ((size_t)b * R.H * R.W + (size_t)pix) * R.D;
  const float *l = ws_all + ((size_t)b * R.ws_rows + i) * R.D;
  float best = __builtin_inff();
  int idx = 0x7fffffff;
  float first = 0.0f;
  for (int d = lane; d < R.D; d += 64) {
    float acc = o[d];
    const float lv = l[d];
    for (int r = 0; r < R.count; ++r) acc = acc + lv;
    if (store_sum) o[d] = acc;
    if (d == 0) first = acc;
    if (idx == 0x7fffffff) idx = d;
    if (acc < best) { best = acc; idx = d; }
  }
  const float m = wave_min_f32(best);
  const int win = wave_min_i32(best == m ? idx : 0x7fffffff);
  first = read_lane(first, 0);
  if (lane == 0) disp_all[(size_t)b * R.H * R.W + (size_t)pix] = (first != first) ? 0 : win;
}

template <int VPL>
int launch_family(const float *cost, const float *image, float *out, float *ws,
                  const SgmShape &P, bool vec, bool first, hipStream_t s) {
  const int waves = P.paths * P.batch;
  const dim3 grid((waves + kSgmBlock / kWave - 1) / (kSgmBlock / kWave)), block(kSgmBlock);
  if (vec && first)
    hipLaunchKernelGGL((sgm_pass_kernel<VPL, true, true>), grid, block, 0, s, cost, image, out, ws, P);
  else if (vec)
    hipLaunchKernelGGL((sgm_pass_kernel<VPL, true, false>), grid, block, 0, s, cost, image, out, ws, P);
  else if (first)
    hipLaunchKernelGGL((sgm_pass_kernel<VPL, false, true>), grid, block, 0, s, cost, image, out, ws, P);
  else
    hipLaunchKernelGGL((sgm_pass_kernel<VPL, false, false>), grid, block, 0, s, cost, image, out, ws, P);
  return check_launch("stm_sgm_f32");
}

int launch_repeat(float *out, const float *ws, int H, int W, int D, int batch, int ws_rows, int x0,
                  int y0, int dx, int dy, int len, int count, hipStream_t s) {
  SgmRepeat R;
  R.H = H; R.W = W; R.D = D; R.batch = batch; R.ws_rows = ws_rows;
  R.x0 = x0; R.y0 = y0; R.dx = dx; R.dy = dy; R.len = len; R.count = count;
  const long n = (long)len * D;
  hipLaunchKernelGGL(sgm_repeat_kernel, dim3((unsigned)((n + 255) / 256), batch), dim3(256), 0, s,
                     out, ws, R);
  return check_launch("stm_sgm_f32 (repeat)");
}

int launch_repeat_wta(float *out, const float *ws, int32_t *disp, int H, int W, int D, int batch,
                      int ws_rows, int x0, int y0, int dx, int dy, int len, int count,
                      int store_sum, hipStream_t s) {
  SgmRepeat R;
  R.H = H; R.W = W; R.D = D; R.batch = batch; R.ws_rows = ws_rows;
  R.x0 = x0; R.y0 = y0; R.dx = dx; R.dy = dy; R.len = len; R.count = count;
  hipLaunchKernelGGL(sgm_repeat_wta_kernel, dim3(len, batch), dim3(64), 0, s, out, ws, disp, R,
                     store_sum);
  return check_launch("stm_sgm_wta_f32 (repeat)");
}

}  // namespace
}  // namespace stm

// Workspace layout (floats): [spill rows: batch x min(H,W) x D, only when a diagonal is
// re-walked] [checkpoints of pair 0] [pair 1] [pair 2] [pair 3]; a pair's block is
// batch x paths x maxseg x D.
struct SgmPlan {
  size_t spill_floats;
  size_t total_floats;
  stm::PairShape P;
};

static int sgm_vpl(int D) {
  const int vpl = (D + stm::kWave - 1) / stm::kWave;
  int VPL = 1;
  while (VPL < vpl) VPL <<= 1;  // 1, 2, 4, 8, 16
  return VPL;
}

static SgmPlan sgm_plan(int H, int W, int D, int batch, bool paired, bool true_paths) {
  SgmPlan plan;
  const stm::SgmQuirk q = stm::sgm_quirk(H, W, true_paths);
  const bool quirk = q.wide_extra > 0 || q.tall_extra > 0;
  plan.spill_floats = quirk ? (size_t)batch * (size_t)(H < W ? H : W) * (size_t)D : 0;
  stm::PairShape &P = plan.P;
  P.H = H; P.W = W; P.D = D; P.batch = batch;
  P.ws_rows = quirk ? (H < W ? H : W) : 0;
  P.tall_extra = q.tall_extra;
  P.pair = 0;
  size_t off = plan.spill_floats;
  const int K = stm::sgm_seg_len(sgm_vpl(D));
  for (int pair = 0; pair < 4; ++pair) {
    const int maxlen = pair == 0 ? W : (pair == 1 ? H : (H < W ? H : W));
    P.paths[pair] = stm::sgm_pair_paths(pair, H, W);
    P.maxseg[pair] = (maxlen + K - 1) / K - 1;
    P.ckpt_off[pair] = (long long)off;
    if (paired) off += (size_t)batch * P.paths[pair] * (size_t)P.maxseg[pair] * (size_t)D;
  }
  plan.total_floats = off;
  return plan;
}

#if !STM_SGM_EXACT
extern "C" size_t stm_sgm_workspace_bytes(int H, int W, int D, int batch, int flags) {
  if (H <= 0 || W <= 0 || D <= 0 || batch <= 0 || D > STM_MAX_DISPARITY) return 0;
  if (stm::sgm_tiled_selected(H, W, D, batch, flags)) {
    // Chosen by the library, not forced: a call whose pointers turn out not to be 16-byte
    // aligned takes the paired form behind the same header, so the workspace fits either.
    const size_t tiled = stm::sgm_tiled_workspace_bytes(H, W, D, batch, flags);
    if (flags & (STM_SGM_TILED | STM_SGM_WALKS | STM_SGM_SKEW)) return tiled;
    const size_t paired = stm::kSgmWorkspaceHeader +
        sgm_plan(H, W, D, batch, true, (flags & STM_SGM_TRUE_PATHS) != 0).total_floats * sizeof(float);
    return tiled > paired ? tiled : paired;
  }
  return sgm_plan(H, W, D, batch, !(flags & STM_SGM_SINGLE_PASSES), (flags & STM_SGM_TRUE_PATHS) != 0).total_floats * sizeof(float);
}

extern "C" int stm_sgm_uses_tiles(int H, int W, int D, int batch, int flags) {
  return stm_sgm_form(H, W, D, batch, flags) != STM_SGM_FORM_PAIRED ? 1 : 0;
}

extern "C" int stm_sgm_form(int H, int W, int D, int batch, int flags) {
  if (H <= 0 || W <= 0 || D <= 0 || batch <= 0 || D > STM_MAX_DISPARITY) return STM_SGM_FORM_PAIRED;
  static_assert(stm::kSgmPaired == STM_SGM_FORM_PAIRED && stm::kSgmTiledSweeps == STM_SGM_FORM_SWEEPS &&
                    stm::kSgmTiledWalks == STM_SGM_FORM_WALKS && stm::kSgmSkew == STM_SGM_FORM_SKEW, "form codes");
  return (int)stm::sgm_form(H, W, D, batch, flags);
}

extern "C" unsigned stm_sgm_status(const void *host_copy_of_first_word) {
  return host_copy_of_first_word ? *static_cast<const unsigned *>(host_copy_of_first_word) : 0u;
}
#endif

namespace stm {
namespace {

template <int VPL>
int launch_ckpt(const float *cost, const float *image, float *ws, const PairShape &P, bool vec,
                hipStream_t s) {
  const long waves = (long)(P.paths[0] + P.paths[1] + P.paths[2] + P.paths[3] +
                            (P.tall_extra > 0 ? 1 : 0)) * P.batch;
  const dim3 grid((unsigned)((waves + kSgmBlock / kWave - 1) / (kSgmBlock / kWave))), block(kSgmBlock);
  if (vec) hipLaunchKernelGGL((sgm_ckpt_kernel<VPL, true>), grid, block, 0, s, cost, image, ws, P);
  else hipLaunchKernelGGL((sgm_ckpt_kernel<VPL, false>), grid, block, 0, s, cost, image, ws, P);
  return check_launch("stm_sgm_f32 (checkpoints)");
}

template <int VPL>
int launch_pair(const float *cost, const float *image, float *out, float *ws, int32_t *disp,
                const PairShape &P, bool vec, bool first, int final_mode, bool nt, hipStream_t s) {
  const long waves = (long)P.paths[P.pair] * P.batch;
  const dim3 grid((unsigned)((waves + kSgmBlock / kWave - 1) / (kSgmBlock / kWave))), block(kSgmBlock);
#define STM_PAIR(VEC_, FIRST_, FINAL_)                                                          \
  do {                                                                                          \
    if (VEC_ && nt)                                                                             \
      hipLaunchKernelGGL((sgm_pair_kernel<VPL, VEC_, FIRST_, FINAL_, VEC_>), grid, block, 0, s, \
                         cost, image, out, ws, disp, P);                                        \
    else                                                                                        \
      hipLaunchKernelGGL((sgm_pair_kernel<VPL, VEC_, FIRST_, FINAL_, false>), grid, block, 0,   \
                         s, cost, image, out, ws, disp, P);                                     \
  } while (0)
  if (final_mode == 1) { if (vec) STM_PAIR(true, false, 1); else STM_PAIR(false, false, 1); }
  else if (final_mode == 2) { if (vec) STM_PAIR(true, false, 2); else STM_PAIR(false, false, 2); }
  else if (vec && first) STM_PAIR(true, true, 0);
  else if (vec) STM_PAIR(true, false, 0);
  else if (first) STM_PAIR(false, true, 0);
  else STM_PAIR(false, false, 0);
#undef STM_PAIR
  return check_launch("stm_sgm_f32 (pair)");
}

}  // namespace
}  // namespace stm

// final_mode: 0 = aggregation only; 1 = aggregation + fused winner-takes-all; 2 = fused
// winner-takes-all without writing the finished sum (out then holds the first six
// directions' sum).
static int sgm_run(const float *cost, const float *left, float *out, int32_t *disp, int H, int W,
                   int D, float penalty1, float penalty2, int flags, void *workspace,
                   size_t workspace_bytes, int batch, void *stream, int final_mode) {
  using namespace stm;
#if !STM_SGM_EXACT
  if (flags & STM_SGM_EXACT_NONFINITE)
    return sgm_run_exact(cost, left, out, disp, H, W, D, penalty1, penalty2, flags, workspace,
                         workspace_bytes, batch, stream, final_mode);
#endif
  STM_REQUIRE(cost && left && out, STM_EINVAL, "stm_sgm_f32: null pointer");
  STM_REQUIRE(H > 0 && W > 0 && D > 0 && batch > 0, STM_EINVAL,
              "stm_sgm_f32: non-positive size (H=%d W=%d D=%d batch=%d)", H, W, D, batch);
  STM_REQUIRE(cost != out, STM_EINVAL, "stm_sgm_f32: cost and out must not alias");
  STM_REQUIRE(D <= STM_MAX_DISPARITY, STM_EUNSUPPORTED,
              "stm_sgm_f32: D=%d exceeds %d", D, STM_MAX_DISPARITY);
  // The reference stores path coordinates as int16 (src/semiglobal.hpp:34-36).
  STM_REQUIRE(H < 32768 && W < 32768, STM_EUNSUPPORTED,
              "stm_sgm_f32: image side exceeds the reference's int16 path coordinates");
  STM_REQUIRE((long long)batch * (W + H) < (1ll << 28) && batch <= 65535, STM_EUNSUPPORTED,
              "stm_sgm_f32: too many paths in one launch");
  const bool paired = !(flags & STM_SGM_SINGLE_PASSES);
  STM_REQUIRE(final_mode == 0 || (paired && disp), STM_EINVAL,
              "stm_sgm_wta_f32: needs a disparity pointer and the paired form");
#if !STM_SGM_EXACT
  if (sgm_tiled_selected(H, W, D, batch, flags)) {
    if ((flags & (STM_SGM_TILED | STM_SGM_WALKS | STM_SGM_SKEW)) || sgm_tiled_aligned(cost, out, workspace, final_mode))
      return sgm_run_tiled(cost, left, out, disp, H, W, D, penalty1, penalty2, flags, workspace,
                           workspace_bytes, batch, as_stream(stream), final_mode);
    // The library's own choice, but this call's pointers are not 16-byte aligned (e.g. an offset
    // view handed in as the output): the paired form takes it, behind the workspace header the
    // caller expects for this shape (its status word reads "ok": the paired form never waits).
    STM_REQUIRE(workspace && workspace_bytes >= kSgmWorkspaceHeader, STM_EWORKSPACE,
                "stm_sgm_f32: needs a workspace of %zu bytes (stm_sgm_workspace_bytes)",
                stm_sgm_workspace_bytes(H, W, D, batch, flags));
    if (hipMemsetAsync(workspace, 0, sizeof(unsigned), as_stream(stream)) != hipSuccess)
      return fail(STM_EHIP, "stm_sgm_f32: hipMemsetAsync failed");
    workspace = static_cast<char *>(workspace) + kSgmWorkspaceHeader;
    workspace_bytes -= kSgmWorkspaceHeader;
  }
#endif
  const bool true_paths = (flags & STM_SGM_TRUE_PATHS) != 0;
  const int dlim = (flags & STM_SGM_FULL_MIN) ? D : D - 1;
  SgmPlan plan = sgm_plan(H, W, D, batch, paired, true_paths);
  const size_t need = plan.total_floats * sizeof(float);
  STM_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), STM_EWORKSPACE,
              "stm_sgm_f32: needs a workspace of %zu bytes (stm_sgm_workspace_bytes)", need);
  float *ws = need ? static_cast<float *>(workspace) : nullptr;
  const int VPL = sgm_vpl(D);
  const uintptr_t align = VPL >= 4 ? 15 : (VPL == 2 ? 7 : 3);
  const bool vec = (D == VPL * kWave) &&
                   ((reinterpret_cast<uintptr_t>(cost) & align) == 0) &&
                   ((reinterpret_cast<uintptr_t>(out) & align) == 0) &&
                   (!ws || (reinterpret_cast<uintptr_t>(ws) & align) == 0);
  hipStream_t s = as_stream(stream);
  const SgmQuirk quirk = sgm_quirk(H, W, true_paths);
  const bool accumulate = (flags & STM_SGM_ACCUMULATE) != 0;
  int rc = STM_OK;

  if (paired) {
    PairShape &P = plan.P;
    P.p1 = penalty1; P.p2 = penalty2;
    P.dlim = dlim; P.true_paths = true_paths ? 1 : 0;
    switch (VPL) {
      case 1: rc = launch_ckpt<1>(cost, left, ws, P, vec, s); break;
      case 2: rc = launch_ckpt<2>(cost, left, ws, P, vec, s); break;
      case 4: rc = launch_ckpt<4>(cost, left, ws, P, vec, s); break;
      case 8: rc = launch_ckpt<8>(cost, left, ws, P, vec, s); break;
      default: rc = launch_ckpt<16>(cost, left, ws, P, vec, s); break;
    }
    if (rc != STM_OK) return rc;
    // Cost + sum beyond the Infinity Cache: stream them past it (see sgm_pair_kernel).
    const bool nt = 2.0 * sizeof(float) * (double)batch * H * W * D > 256.0 * 1024 * 1024;
    for (int pair = 0; pair < 4; ++pair) {
      P.pair = pair;
      const bool first = pair == 0 && !accumulate;
      // The fused reduction belongs to the last pair; on wide images the re-walked
      // anti-diagonal still receives adds afterwards and is reduced by the repeat kernel.
      const int fm = pair == 3 ? final_mode : 0;
      // Tall images: the corner diagonal's early re-walks precede the (1,1) forward pass.
      if (pair == 2 && quirk.tall_extra > 0) {
        rc = launch_repeat(out, ws, H, W, D, batch, P.ws_rows, W - 1, H - 1, -1, -1, W, quirk.tall_extra, s);
        if (rc != STM_OK) return rc;
      }
      switch (VPL) {
        case 1: rc = launch_pair<1>(cost, left, out, ws, disp, P, vec, first, fm, nt, s); break;
        case 2: rc = launch_pair<2>(cost, left, out, ws, disp, P, vec, first, fm, nt, s); break;
        case 4: rc = launch_pair<4>(cost, left, out, ws, disp, P, vec, first, fm, nt, s); break;
        case 8: rc = launch_pair<8>(cost, left, out, ws, disp, P, vec, first, fm, nt, s); break;
        default: rc = launch_pair<16>(cost, left, out, ws, disp, P, vec, first, fm, nt, s); break;
      }
      if (rc != STM_OK) return rc;
      if (pair == 2 && quirk.wide_extra > 0)
        rc = launch_repeat(out, ws, H, W, D, batch, P.ws_rows, W - 1, H - 1, -1, -1, H, quirk.wide_extra, s);
      else if (pair == 3 && quirk.wide_extra > 0 && final_mode == 0)
        rc = launch_repeat(out, ws, H, W, D, batch, P.ws_rows, 0, H - 1, 1, -1, H, quirk.wide_extra, s);
      else if (pair == 3 && quirk.wide_extra > 0)
        rc = launch_repeat_wta(out, ws, disp, H, W, D, batch, P.ws_rows, 0, H - 1, 1, -1, H,
                               quirk.wide_extra, final_mode == 1 ? 1 : 0, s);
      if (rc != STM_OK) return rc;
    }
    return STM_OK;
  }

  SgmShape P;
  P.H = H; P.W = W; P.D = D; P.p1 = penalty1; P.p2 = penalty2; P.batch = batch;
  P.ws_rows = plan.P.ws_rows;
  P.dlim = dlim; P.true_paths = true_paths ? 1 : 0;
  for (int family = 0; family < 8; ++family) {
    P.family = family;
    P.paths = sgm_family_paths(family, H, W, true_paths);
    const bool first = family == 0 && !accumulate;
    switch (VPL) {
      case 1: rc = launch_family<1>(cost, left, out, ws, P, vec, first, s); break;
      case 2: rc = launch_family<2>(cost, left, out, ws, P, vec, first, s); break;
      case 4: rc = launch_family<4>(cost, left, out, ws, P, vec, first, s); break;
      case 8: rc = launch_family<8>(cost, left, out, ws, P, vec, first, s); break;
      default: rc = launch_family<16>(cost, left, out, ws, P, vec, first, s); break;
    }
    if (rc != STM_OK) return rc;
    // Re-walked diagonals (Q6): the repeated adds go where the table order puts them.
    if (family == kUp && quirk.tall_extra > 0)
      rc = launch_repeat(out, ws, P.H, P.W, P.D, P.batch, P.ws_rows, W - 1, H - 1, -1, -1, W, quirk.tall_extra, s);
    else if (family == kUpLeft && quirk.wide_extra > 0)
      rc = launch_repeat(out, ws, P.H, P.W, P.D, P.batch, P.ws_rows, W - 1, H - 1, -1, -1, H, quirk.wide_extra, s);
    else if (family == kUpRight && quirk.wide_extra > 0)
      rc = launch_repeat(out, ws, P.H, P.W, P.D, P.batch, P.ws_rows, 0, H - 1, 1, -1, H, quirk.wide_extra, s);
    if (rc != STM_OK) return rc;
  }
  return STM_OK;
}

#if STM_SGM_EXACT
int stm::sgm_run_exact(const float *cost, const float *left, float *out, int32_t *disp, int H,
                       int W, int D, float penalty1, float penalty2, int flags, void *workspace,
                       size_t workspace_bytes, int batch, void *stream, int final_mode) {
  return sgm_run(cost, left, out, disp, H, W, D, penalty1, penalty2, flags, workspace,
                 workspace_bytes, batch, stream, final_mode);
}
#else
extern "C" int stm_sgm_f32(const float *cost, const float *left, float *out, int H, int W,
                           int D, float penalty1, float penalty2, int flags, void *workspace,
                           size_t workspace_bytes, int batch, void *stream) {
  return sgm_run(cost, left, out, nullptr, H, W, D, penalty1, penalty2, flags, workspace,
                 workspace_bytes, batch, stream, 0);
}

extern "C" int stm_sgm_wta_f32(const float *cost, const float *left, float *sum,
                               int32_t *disparity, int H, int W, int D, float penalty1,
                               float penalty2, int flags, void *workspace, size_t workspace_bytes,
                               int batch, void *stream) {
  STM_REQUIRE(disparity, STM_EINVAL, "stm_sgm_wta_f32: null disparity pointer");
  STM_REQUIRE(!(flags & STM_SGM_SINGLE_PASSES), STM_EINVAL,
              "stm_sgm_wta_f32: the fused reduction exists in the paired form only");
  return sgm_run(cost, left, sum, disparity, H, W, D, penalty1, penalty2, flags, workspace,
                 workspace_bytes, batch, stream, (flags & STM_SGM_KEEP_SUM) ? 1 : 2);
}
#endif  // !STM_SGM_EXACT
