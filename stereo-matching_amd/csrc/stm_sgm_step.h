// One step of the semiglobal recurrence for a wave that holds a pixel's disparities in its
// lanes (VPL consecutive values per lane), shared by every aggregation kernel.
//
// Reference: src/semiglobal.cpp:129-157 (SGMCostOperator::operator(), the loop body).
#pragma once

#include "stm_common.h"

#ifndef STM_SGM_EXACT
#define STM_SGM_EXACT 0
#endif

namespace stm {
namespace {

constexpr bool kExactNonFinite = STM_SGM_EXACT != 0;

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

// P2' of the step between two neighbouring pixels of a path (semiglobal.cpp:137-139):
// max(P1, P2 / |I(p_i) - I(p_i-1)|).  |a - b| is exactly |b - a|, so a forward and an
// inverse pass crossing the same pixel boundary use the same value.
__device__ __forceinline__ float sgm_penalty(float inten, float prev_i, float p1, float p2) {
  const float q = p2 / fabsf(inten - prev_i);
  return p1 < q ? q : p1;  // std::max(p1, q)
}

// A copy of a wave-uniform float that lives in a VGPR (DPP operands must).
__device__ __forceinline__ float in_vgpr(float x) {
  asm volatile("" : "+v"(x));
  return x;
}

// Per-wave constants and scratch of sgm_step.
struct StepCtx {
  float p1, p1v;     // P1 as a scalar and as a VGPR copy (DPP operands must be VGPRs)
  float below, above;  // DPP results; lane 0 of `below` / lane 63 of `above` stay +inf
  int d0, dlim;
  bool drop_tail;    // VEC only (D = 64 * VPL): this lane's last element is d = D-1 and the
                     // reference minimum leaves it out
  __device__ __forceinline__ void init(float p1_, int d0_, int dlim_, bool drop_tail_) {
    p1 = p1_;
    p1v = in_vgpr(p1_);
    below = above = in_vgpr(__builtin_inff());
    d0 = d0_;
    dlim = dlim_;
    drop_tail = drop_tail_;
  }
};

// One step of the recurrence, pixel i-1 -> pixel i of a path (semiglobal.cpp:129-157).
// p2a: sgm_penalty of the step (wave-uniform).
template <int VPL, bool VEC>
__device__ __forceinline__ Row<VPL> sgm_step(const Row<VPL> &prev, const Row<VPL> &c, float p2a,
                                             StepCtx &X) {
  const float inf = __builtin_inff();
  // m over d in [0, dlim): dlim = D-1 reproduces the reference, whose minimum leaves the
  // element d = D-1 out (Q5); dlim = D is the corrected full-range minimum.
  float lm;
  if constexpr (VEC) {
    lm = X.drop_tail ? inf : prev.v[VPL - 1];
#pragma unroll
    for (int j = 0; j + 1 < VPL - 1; j += 2) lm = vmin3_f32(lm, prev.v[j], prev.v[j + 1]);
    if constexpr (VPL % 2 == 0) lm = vmin_f32(lm, prev.v[VPL - 2]);
  } else {
    lm = inf;
#pragma unroll
    for (int j = 0; j < VPL; ++j)
      if (X.d0 + j < X.dlim) lm = vmin_f32(lm, prev.v[j]);
  }
  float m = wave_min_f32(lm);
  if constexpr (!VEC) {
    if (X.dlim == 0) m = read_lane(prev.v[0], 0);  // D = 1: empty range, *end() is element D-1
  }
  if constexpr (kExactNonFinite) {
    // std::min_element (cpp:130) is a strict-'<' scan that starts at element 0: a NaN there
    // is never replaced, a NaN further on never wins.  v_min_f32 ignores NaN operands, which
    // is the second half of that rule; the first half is this select.
    const float head = read_lane(prev.v[0], 0);
    m = (head != head) ? head : m;
  }
  const float far = m + p2a;
  lane_up_add(X.below, prev.v[VPL - 1], X.p1v);  // L(d0-1) + P1
  lane_down_add(X.above, prev.v[0], X.p1v);      // L(d0+VPL) + P1
  float w[VPL];
#pragma unroll
  for (int j = 0; j < VPL; ++j) w[j] = prev.v[j] + X.p1;
  Row<VPL> cur;
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const float lo = j == 0 ? X.below : w[j - 1];
    const float hi = j == VPL - 1 ? X.above : w[j + 1];
    // get_min (cpp:11-16) is a strict-'<' chain; as a value that is the minimum of the
    // four candidates (v_min3_f32 + v_min_f32).
    float t = vmin_f32(vmin3_f32(prev.v[j], lo, hi), far);
    // get_min starts from L(d) and replaces it by a smaller candidate: a NaN L(d) stays.
    if constexpr (kExactNonFinite) t = (prev.v[j] != prev.v[j]) ? prev.v[j] : t;
    const float a = c.v[j] + t;
    cur.v[j] = a - m;
  }
  return cur;
}


// ---------------------------------------------------------------------------------
// N independent steps at once (N = 2 or 3; whole lanes only, D = 64 * VPL).
//
// A wave issues in order, and a DPP instruction may read a VGPR only two wait states after
// the VALU write that produced it, so the six-stage wave minimum of one step spends as many
// issue slots on s_nop as on v_min.  Steps of different paths are independent: interleaved
// stage by stage, each chain's next DPP instruction is separated from its producer by the
// other chains' instructions and (for N = 3) no padding is left at all.  Same instructions
// per chain, same results.
// ---------------------------------------------------------------------------------
template <int N>
struct StepCtxN {
  float p1, p1v;
  float below[N], above[N];  // lane 0 of below[k] / lane 63 of above[k] stay +inf
  bool drop_tail;
  __device__ __forceinline__ void init(float p1_, bool drop_tail_) {
    p1 = p1_;
    p1v = in_vgpr(p1_);
#pragma unroll
    for (int k = 0; k < N; ++k) below[k] = above[k] = in_vgpr(__builtin_inff());
    drop_tail = drop_tail_;
  }
};

#define STM_DPP_MIN(reg, ctrl) "v_min_f32_dpp " reg ", " reg ", " reg " " ctrl "\n\t"
#define STM_ROW_SHR1 "row_shr:1 row_mask:0xf bank_mask:0xf"
#define STM_ROW_SHR2 "row_shr:2 row_mask:0xf bank_mask:0xf"
#define STM_ROW_SHR4 "row_shr:4 row_mask:0xf bank_mask:0xf"
#define STM_ROW_SHR8 "row_shr:8 row_mask:0xf bank_mask:0xf"
#define STM_BCAST15 "row_bcast:15 row_mask:0xa bank_mask:0xf"
#define STM_BCAST31 "row_bcast:31 row_mask:0xc bank_mask:0xf"

// In-place wave minima of two / three registers (lane 63 holds the result, as in
// wave_min_f32).  The leading s_nop covers a VALU write of the inputs just before.
__device__ __forceinline__ void wave_min2_f32(float &a, float &b) {
  asm("s_nop 1\n\t"
      STM_DPP_MIN("%0", STM_ROW_SHR1) STM_DPP_MIN("%1", STM_ROW_SHR1) "s_nop 0\n\t"
      STM_DPP_MIN("%0", STM_ROW_SHR2) STM_DPP_MIN("%1", STM_ROW_SHR2) "s_nop 0\n\t"
      STM_DPP_MIN("%0", STM_ROW_SHR4) STM_DPP_MIN("%1", STM_ROW_SHR4) "s_nop 0\n\t"
      STM_DPP_MIN("%0", STM_ROW_SHR8) STM_DPP_MIN("%1", STM_ROW_SHR8) "s_nop 0\n\t"
      STM_DPP_MIN("%0", STM_BCAST15) STM_DPP_MIN("%1", STM_BCAST15) "s_nop 0\n\t"
      STM_DPP_MIN("%0", STM_BCAST31) "v_min_f32_dpp %1, %1, %1 " STM_BCAST31
      : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void wave_min3_f32(float &a, float &b, float &c) {
  asm("s_nop 1\n\t"
      STM_DPP_MIN("%0", STM_ROW_SHR1) STM_DPP_MIN("%1", STM_ROW_SHR1) STM_DPP_MIN("%2", STM_ROW_SHR1)
      STM_DPP_MIN("%0", STM_ROW_SHR2) STM_DPP_MIN("%1", STM_ROW_SHR2) STM_DPP_MIN("%2", STM_ROW_SHR2)
      STM_DPP_MIN("%0", STM_ROW_SHR4) STM_DPP_MIN("%1", STM_ROW_SHR4) STM_DPP_MIN("%2", STM_ROW_SHR4)
      STM_DPP_MIN("%0", STM_ROW_SHR8) STM_DPP_MIN("%1", STM_ROW_SHR8) STM_DPP_MIN("%2", STM_ROW_SHR8)
      STM_DPP_MIN("%0", STM_BCAST15) STM_DPP_MIN("%1", STM_BCAST15) STM_DPP_MIN("%2", STM_BCAST15)
      STM_DPP_MIN("%0", STM_BCAST31) STM_DPP_MIN("%1", STM_BCAST31) "v_min_f32_dpp %2, %2, %2 " STM_BCAST31
      : "+v"(a), "+v"(b), "+v"(c));
}

// below[k] = (value of lane - 1 of lo[k]) + p1, above[k] = (value of lane + 1 of hi[k]) + p1
// for all chains with one leading s_nop (the sources are old registers; the nop covers p1v).
template <int N>
__device__ __forceinline__ void lane_shift_adds(float (&below)[N], float (&above)[N], const float (&lo)[N],
                                                const float (&hi)[N], float p1v) {
  if constexpr (N == 2) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %5, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %6, %8 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %7, %8 wave_shl:1 row_mask:0xf bank_mask:0xf"
        : "+v"(below[0]), "+v"(below[1]), "+v"(above[0]), "+v"(above[1])
        : "v"(lo[0]), "v"(lo[1]), "v"(hi[0]), "v"(hi[1]), "v"(p1v));
  } else {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %6, %12 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %7, %12 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %8, %12 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %9, %12 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %10, %12 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %5, %11, %12 wave_shl:1 row_mask:0xf bank_mask:0xf"
        : "+v"(below[0]), "+v"(below[1]), "+v"(below[2]), "+v"(above[0]), "+v"(above[1]), "+v"(above[2])
        : "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(p1v));
  }
}

// cur[k] = one step (semiglobal.cpp:129-157) from prev[k] into a pixel with costs c[k] and
// penalty p2a[k], for N independent paths.
template <int VPL, int N>
__device__ __forceinline__ void sgm_step_n(const Row<VPL> (&prev)[N], const Row<VPL> (&c)[N],
                                           const float (&p2a)[N], StepCtxN<N> &X, Row<VPL> (&cur)[N]) {
  static_assert(N == 2 || N == 3, "two or three chains");
  const float inf = __builtin_inff();
  float lm[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    lm[k] = X.drop_tail ? inf : prev[k].v[VPL - 1];
#pragma unroll
    for (int j = 0; j + 1 < VPL - 1; j += 2) lm[k] = vmin3_f32(lm[k], prev[k].v[j], prev[k].v[j + 1]);
    if constexpr (VPL % 2 == 0) lm[k] = vmin_f32(lm[k], prev[k].v[VPL - 2]);
  }
  if constexpr (N == 2) wave_min2_f32(lm[0], lm[1]);
  else wave_min3_f32(lm[0], lm[1], lm[2]);
  float m[N], far[N], lo[N], hi[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    m[k] = read_lane(lm[k], 63);
    far[k] = m[k] + p2a[k];
    lo[k] = prev[k].v[VPL - 1];
    hi[k] = prev[k].v[0];
  }
  lane_shift_adds<N>(X.below, X.above, lo, hi, X.p1v);
#pragma unroll
  for (int k = 0; k < N; ++k) {
    float w[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) w[j] = prev[k].v[j] + X.p1;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      const float l = j == 0 ? X.below[k] : w[j - 1];
      const float h = j == VPL - 1 ? X.above[k] : w[j + 1];
      const float t = vmin_f32(vmin3_f32(prev[k].v[j], l, h), far[k]);
      const float a = c[k].v[j] + t;
      cur[k].v[j] = a - m[k];
    }
  }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------
// Two pixels per wave (D = 128): lanes 0-31 hold one pixel's disparities, four per lane, lanes
// 32-63 another pixel's, and one instruction sequence steps both paths.
//
// The kernels that run at D <= 128 are bound by VALU instructions issued (a wave64 instruction
// holds its SIMD for four cycles whatever it computes).  With two values per lane and one
// pixel per wave a step is ~21 of them, 8 for the six-stage wave minimum and its broadcast
// alone; here one sequence of ~27 serves two pixels: the minimum is a five-stage butterfly
// over 32 lanes that leaves the result in every lane of its half (no v_readlane, no copy
// back), and the adds come packed.  Same operations on the same operands per pixel -> same
// bits.
// ---------------------------------------------------------------------------------
struct StepCtxH2 {
  float p1v;           // P1 in a VGPR
  float below, above;  // lanes 0 / 32 of `below` and 31 / 63 of `above` stay +inf
  bool drop_tail;      // this lane's last element is d = D-1 and the reference minimum leaves it out (Q5)
  __device__ __forceinline__ void init(float p1, int hl, int dlim) {
    p1v = in_vgpr(p1);
    below = above = in_vgpr(__builtin_inff());
    drop_tail = hl == 31 && dlim < 128;
  }
};

// Minimum over the 32 lanes of the lane's half, in every lane: xor 1, xor 2 (quad
// permutations), mirror within 8, mirror within 16, then the two rows of a half are exchanged
// (v_permlane16_swap of two copies: one ends up holding the even rows' values in both rows of
// a pair, the other the odd rows').  NaNs are ignored unless every lane holds one, as in
// wave_min_f32.
__device__ __forceinline__ float half_min_f32(float v) {
  float t;
  asm("s_nop 1\n\t"
      "v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_mov_b32 %1, %0\n\t"
      "s_nop 1\n\t"
      "v_permlane16_swap_b32 %1, %0\n\t"
      "s_nop 1\n\t"
      "v_min_f32 %0, %0, %1"
      : "+v"(v), "=&v"(t));
  return v;
}

// Two such minima at once (TYPE: f32, or i32 for the winner's index in the fused
// winner-takes-all), interleaved so that one's wait states are the other's instructions; a leading s_nop covers a VALU write of the inputs just before.
#define STM_HALF_MIN2(TYPE, a, b)                                                                     \
  do {                                                                                                \
    unsigned ta_, tb_;                                                                                \
    asm("s_nop 1\n\t"                                                                                 \
        "v_min_" TYPE "_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"            \
        "v_min_" TYPE "_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"            \
        "s_nop 0\n\t"                                                                                 \
        "v_min_" TYPE "_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"            \
        "v_min_" TYPE "_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"            \
        "s_nop 0\n\t"                                                                                 \
        "v_min_" TYPE "_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"                \
        "v_min_" TYPE "_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"                \
        "s_nop 0\n\t"                                                                                 \
        "v_min_" TYPE "_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"                     \
        "v_min_" TYPE "_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"                     \
        "s_nop 0\n\t"                                                                                 \
        "v_mov_b32 %2, %0\n\t"                                                                        \
        "v_mov_b32 %3, %1\n\t"                                                                        \
        "s_nop 0\n\t"                                                                                 \
        "v_permlane16_swap_b32 %2, %0\n\t"                                                            \
        "v_permlane16_swap_b32 %3, %1\n\t"                                                            \
        "s_nop 0\n\t"                                                                                 \
        "v_min_" TYPE " %0, %0, %2\n\t"                                                               \
        "v_min_" TYPE " %1, %1, %3"                                                                    \
        : "+v"(a), "+v"(b), "=&v"(ta_), "=&v"(tb_));                                                  \
  } while (0)
__device__ __forceinline__ void half_min2_f32(float &a, float &b) { STM_HALF_MIN2("f32", a, b); }
__device__ __forceinline__ void half_min2_i32(int &a, int &b) { STM_HALF_MIN2("i32", a, b); }
#undef STM_HALF_MIN2

// One step for both halves (semiglobal.cpp:129-157).  `pen` holds the penalties P2' of the
// lane's half, one per lane of each 16-lane row (both rows of a half hold the same sixteen);
// PEN_LANE selects this step's.
template <int PEN_LANE>
__device__ __forceinline__ Row<4> sgm_step_h2(const Row<4> &prev, const Row<4> &c, float pen, StepCtxH2 &X) {
  static_assert(!kExactNonFinite, "the exact non-finite variant uses sgm_step");
  const float inf = __builtin_inff();
  float lm = X.drop_tail ? inf : prev.v[3];
  lm = vmin_f32(vmin3_f32(lm, prev.v[0], prev.v[1]), prev.v[2]);
  const float m = half_min_f32(lm);
  float far;  // m + P2'
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
      : "=v"(far) : "v"(pen), "v"(m), "n"(PEN_LANE));
  // L(d0-1) + P1 from the lane below, L(d0+4) + P1 from the lane above, inside each half: a
  // shift inside the 16-lane rows, then the lanes at a row boundary inside a half (16 and 48 /
  // 15 and 47) by a wave shift restricted to those rows and banks (the other lanes of the
  // bank receive the value they already hold).  Lanes 0 / 32 and 31 / 63 are never written.
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %2, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %3, %4 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %2, %4 wave_shr:1 row_mask:0xa bank_mask:0x1\n\t"
      "v_add_f32_dpp %1, %3, %4 wave_shl:1 row_mask:0x5 bank_mask:0x8"
      : "+v"(X.below), "+v"(X.above) : "v"(prev.v[3]), "v"(prev.v[0]), "v"(X.p1v));
  const f32x2 p1 = {X.p1v, X.p1v};
  const f32x2 w01 = f32x2{prev.v[0], prev.v[1]} + p1;
  const f32x2 w23 = f32x2{prev.v[2], prev.v[3]} + p1;
  float t[4];
  t[0] = vmin_f32(vmin3_f32(prev.v[0], X.below, w01.y), far);
  t[1] = vmin_f32(vmin3_f32(prev.v[1], w01.x, w23.x), far);
  t[2] = vmin_f32(vmin3_f32(prev.v[2], w01.y, w23.y), far);
  t[3] = vmin_f32(vmin3_f32(prev.v[3], w23.x, X.above), far);
  const f32x2 m2 = {m, m};
  const f32x2 r01 = (f32x2{c.v[0], c.v[1]} + f32x2{t[0], t[1]}) - m2;
  const f32x2 r23 = (f32x2{c.v[2], c.v[3]} + f32x2{t[2], t[3]}) - m2;
  Row<4> cur;
  cur.v[0] = r01.x; cur.v[1] = r01.y; cur.v[2] = r23.x; cur.v[3] = r23.y;
  return cur;
}

// Two independent steps of the two-pixel form at once (the forward and the inverse chain of
// the same lines).  A wave issues in order; a DPP instruction may read a register only two
// wait states after the VALU write that produced it, and the butterfly minimum is a chain of
// such reads.  The two chains' cross-lane instructions -- minima, neighbour adds, the P2'
// broadcast -- are interleaved here so that each one's wait states are the other's
// instructions: 24 instructions, no padding.  What follows per chain (packed adds, three-way
// minima) has no cross-lane reads and is left to the compiler's scheduler.
template <int PEN_F, int PEN_I>
__device__ __forceinline__ void sgm_step_h2x2(Row<4> &pf, const Row<4> &cf, Row<4> &pi, const Row<4> &ci,
                                              float pen, StepCtxH2 &XF, StepCtxH2 &XI) {
  static_assert(!kExactNonFinite, "the exact non-finite variant uses sgm_step");
  const float inf = __builtin_inff();
  float mf = XF.drop_tail ? inf : pf.v[3];
  float mi = XI.drop_tail ? inf : pi.v[3];
  mf = vmin_f32(vmin3_f32(mf, pf.v[0], pf.v[1]), pf.v[2]);
  mi = vmin_f32(vmin3_f32(mi, pi.v[0], pi.v[1]), pi.v[2]);
  float tf, ti, farf, fari;
#define STM_H2_MIN(reg, ctrl) "v_min_f32_dpp " reg ", " reg ", " reg " " ctrl " row_mask:0xf bank_mask:0xf\n\t"
  asm("s_nop 1\n\t"
      STM_H2_MIN("%0", "quad_perm:[1,0,3,2]")
      STM_H2_MIN("%1", "quad_perm:[1,0,3,2]")
      "v_add_f32_dpp %6, %10, %15 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      STM_H2_MIN("%0", "quad_perm:[2,3,0,1]")
      STM_H2_MIN("%1", "quad_perm:[2,3,0,1]")
      "v_add_f32_dpp %7, %11, %15 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
      STM_H2_MIN("%0", "row_half_mirror")
      STM_H2_MIN("%1", "row_half_mirror")
      "v_add_f32_dpp %8, %12, %15 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      STM_H2_MIN("%0", "row_mirror")
      STM_H2_MIN("%1", "row_mirror")
      "v_add_f32_dpp %9, %13, %15 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32 %2, %0\n\t"
      "v_mov_b32 %3, %1\n\t"
      "v_add_f32_dpp %6, %10, %15 wave_shr:1 row_mask:0xa bank_mask:0x1\n\t"
      "v_permlane16_swap_b32 %2, %0\n\t"
      "v_permlane16_swap_b32 %3, %1\n\t"
      "v_add_f32_dpp %7, %11, %15 wave_shl:1 row_mask:0x5 bank_mask:0x8\n\t"
      "v_add_f32_dpp %8, %12, %15 wave_shr:1 row_mask:0xa bank_mask:0x1\n\t"
      "v_min_f32 %0, %0, %2\n\t"
      "v_min_f32 %1, %1, %3\n\t"
      "v_add_f32_dpp %9, %13, %15 wave_shl:1 row_mask:0x5 bank_mask:0x8\n\t"
      "v_add_f32_dpp %4, %14, %0 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %14, %1 row_newbcast:%17 row_mask:0xf bank_mask:0xf"
      : "+v"(mf), "+v"(mi), "=&v"(tf), "=&v"(ti), "=&v"(farf), "=&v"(fari), "+v"(XF.below), "+v"(XF.above),
        "+v"(XI.below), "+v"(XI.above)
      : "v"(pf.v[3]), "v"(pf.v[0]), "v"(pi.v[3]), "v"(pi.v[0]), "v"(pen), "v"(XF.p1v), "n"(PEN_F), "n"(PEN_I));
#undef STM_H2_MIN
  auto tail = [](Row<4> &prev, const Row<4> &c, float m, float far, float below, float above, float p1v) {
    const f32x2 p1 = {p1v, p1v};
    const f32x2 w01 = f32x2{prev.v[0], prev.v[1]} + p1;
    const f32x2 w23 = f32x2{prev.v[2], prev.v[3]} + p1;
    float t[4];
    t[0] = vmin_f32(vmin3_f32(prev.v[0], below, w01.y), far);
    t[1] = vmin_f32(vmin3_f32(prev.v[1], w01.x, w23.x), far);
    t[2] = vmin_f32(vmin3_f32(prev.v[2], w01.y, w23.y), far);
    t[3] = vmin_f32(vmin3_f32(prev.v[3], w23.x, above), far);
    const f32x2 m2 = {m, m};
    const f32x2 r01 = (f32x2{c.v[0], c.v[1]} + f32x2{t[0], t[1]}) - m2;
    const f32x2 r23 = (f32x2{c.v[2], c.v[3]} + f32x2{t[2], t[3]}) - m2;
    prev.v[0] = r01.x; prev.v[1] = r01.y; prev.v[2] = r23.x; prev.v[3] = r23.y;
  };
  tail(pf, cf, mf, farf, XF.below, XF.above, XF.p1v);
  tail(pi, ci, mi, fari, XI.below, XI.above, XF.p1v);
}

// Three independent steps of the two-pixel form at once (the three directions of a row sweep):
// as sgm_step_h2x2, 36 cross-lane instructions without padding.
template <int PEN0, int PEN1, int PEN2>
__device__ __forceinline__ void sgm_step_h2x3(Row<4> (&p)[3], const Row<4> (&c)[3], float pen, float p1v,
                                              float (&below)[3], float (&above)[3], bool drop_tail) {
  static_assert(!kExactNonFinite, "the exact non-finite variant uses sgm_step");
  const float inf = __builtin_inff();
  float m[3], t[3], far[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    m[k] = drop_tail ? inf : p[k].v[3];
    m[k] = vmin_f32(vmin3_f32(m[k], p[k].v[0], p[k].v[1]), p[k].v[2]);
  }
#define STM_H2_MIN3(ctrl)                                                              \
  "v_min_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                    \
  "v_min_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                    \
  "v_min_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf\n\t"
  asm("s_nop 1\n\t"
      STM_H2_MIN3("quad_perm:[1,0,3,2]")
      STM_H2_MIN3("quad_perm:[2,3,0,1]")
      STM_H2_MIN3("row_half_mirror")
      STM_H2_MIN3("row_mirror")
      "v_add_f32_dpp %9, %15, %22 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %10, %16, %22 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %11, %17, %22 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %12, %18, %22 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %13, %19, %22 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %14, %20, %22 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32 %3, %0\n\t"
      "v_mov_b32 %4, %1\n\t"
      "v_mov_b32 %5, %2\n\t"
      "v_add_f32_dpp %9, %15, %22 wave_shr:1 row_mask:0xa bank_mask:0x1\n\t"
      "v_add_f32_dpp %10, %16, %22 wave_shr:1 row_mask:0xa bank_mask:0x1\n\t"
      "v_add_f32_dpp %11, %17, %22 wave_shr:1 row_mask:0xa bank_mask:0x1\n\t"
      "v_permlane16_swap_b32 %3, %0\n\t"
      "v_permlane16_swap_b32 %4, %1\n\t"
      "v_permlane16_swap_b32 %5, %2\n\t"
      "v_add_f32_dpp %12, %18, %22 wave_shl:1 row_mask:0x5 bank_mask:0x8\n\t"
      "v_add_f32_dpp %13, %19, %22 wave_shl:1 row_mask:0x5 bank_mask:0x8\n\t"
      "v_add_f32_dpp %14, %20, %22 wave_shl:1 row_mask:0x5 bank_mask:0x8\n\t"
      "v_min_f32 %0, %0, %3\n\t"
      "v_min_f32 %1, %1, %4\n\t"
      "v_min_f32 %2, %2, %5\n\t"
      "v_add_f32_dpp %6, %21, %0 row_newbcast:%23 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %21, %1 row_newbcast:%24 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %8, %21, %2 row_newbcast:%25 row_mask:0xf bank_mask:0xf"
      : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(far[0]), "=&v"(far[1]),
        "=&v"(far[2]), "+v"(below[0]), "+v"(below[1]), "+v"(below[2]), "+v"(above[0]), "+v"(above[1]),
        "+v"(above[2])
      : "v"(p[0].v[3]), "v"(p[1].v[3]), "v"(p[2].v[3]), "v"(p[0].v[0]), "v"(p[1].v[0]), "v"(p[2].v[0]), "v"(pen),
        "v"(p1v), "n"(PEN0), "n"(PEN1), "n"(PEN2));
#undef STM_H2_MIN3
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const f32x2 p1 = {p1v, p1v};
    const f32x2 w01 = f32x2{p[k].v[0], p[k].v[1]} + p1;
    const f32x2 w23 = f32x2{p[k].v[2], p[k].v[3]} + p1;
    float u[4];
    u[0] = vmin_f32(vmin3_f32(p[k].v[0], below[k], w01.y), far[k]);
    u[1] = vmin_f32(vmin3_f32(p[k].v[1], w01.x, w23.x), far[k]);
    u[2] = vmin_f32(vmin3_f32(p[k].v[2], w01.y, w23.y), far[k]);
    u[3] = vmin_f32(vmin3_f32(p[k].v[3], w23.x, above[k]), far[k]);
    const f32x2 m2 = {m[k], m[k]};
    const f32x2 r01 = (f32x2{c[k].v[0], c[k].v[1]} + f32x2{u[0], u[1]}) - m2;
    const f32x2 r23 = (f32x2{c[k].v[2], c[k].v[3]} + f32x2{u[2], u[3]}) - m2;
    p[k].v[0] = r01.x; p[k].v[1] = r01.y; p[k].v[2] = r23.x; p[k].v[3] = r23.y;
  }
}

// ---------------------------------------------------------------------------------
// One step with its front half scheduled by hand (whole lanes only; VPL = 2 or 4).
//
// In sgm_step every DPP instruction of the six-stage wave minimum sits behind an s_nop 1 (two
// wait states after the VALU write of its source), and so do the two cross-lane adds: 8 of
// the step's ~50 issue slots.  Here the slots between the minimum's stages carry the step's
// own independent work instead -- L(d) + P1 for the lane's values, the two cross-lane adds
// and the three-way minima that do not need the wave minimum yet -- so only the last stage
// still waits.  Same instructions, same operands, same results; about a third fewer issue
// slots, which is what bounds the kernels that use it.
// ---------------------------------------------------------------------------------
template <int VPL>
__device__ __forceinline__ Row<VPL> sgm_step_fused(const Row<VPL> &prev, const Row<VPL> &c, float p2a,
                                                   StepCtx &X) {
  static_assert(VPL == 2 || VPL == 4, "hand-scheduled for two or four values per lane");
  const float inf = __builtin_inff();
  const float lmt = X.drop_tail ? inf : prev.v[VPL - 1];  // Q5: lane 63 leaves d = D-1 out
  float t[VPL], m;
  if constexpr (VPL == 4) {
    float lm, w0, w1, w2, w3;
    asm("v_min3_f32 %0, %15, %11, %12\n\t"
        "v_min_f32 %0, %0, %13\n\t"
        "v_add_f32 %1, %16, %11\n\t"
        "v_add_f32 %2, %16, %12\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32 %3, %16, %13\n\t"
        "v_add_f32 %4, %16, %14\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %9, %14, %17 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %10, %11, %17 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_min3_f32 %6, %12, %1, %3\n\t"
        "v_min3_f32 %7, %13, %2, %4\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_min3_f32 %5, %11, %9, %2\n\t"
        "v_min3_f32 %8, %14, %3, %10\n\t"
        "v_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "=&v"(lm), "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]),
          "=&v"(t[3]), "+v"(X.below), "+v"(X.above)
        : "v"(prev.v[0]), "v"(prev.v[1]), "v"(prev.v[2]), "v"(prev.v[3]), "v"(lmt), "s"(X.p1), "v"(X.p1v));
    m = read_lane(lm, 63);
  } else {
    float lm, w0, w1;
    asm("v_min_f32 %0, %9, %7\n\t"
        "v_add_f32 %1, %10, %7\n\t"
        "v_add_f32 %2, %10, %8\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %5, %8, %11 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %6, %7, %11 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_min3_f32 %3, %7, %5, %2\n\t"
        "v_min3_f32 %4, %8, %1, %6\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "=&v"(lm), "=&v"(w0), "=&v"(w1), "=&v"(t[0]), "=&v"(t[1]), "+v"(X.below), "+v"(X.above)
        : "v"(prev.v[0]), "v"(prev.v[1]), "v"(lmt), "s"(X.p1), "v"(X.p1v));
    m = read_lane(lm, 63);
  }
  const float far = m + p2a;
  Row<VPL> cur;
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const float a = c.v[j] + vmin_f32(t[j], far);
    cur.v[j] = a - m;
  }
  return cur;
}

}  // namespace
}  // namespace stm
