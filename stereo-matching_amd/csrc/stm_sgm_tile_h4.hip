// Semiglobal aggregation, tiled form: the tile kernel for D = 128 with FOUR pixels per wave (see
// stm_sgm_tiled.hip for the whole form; sgm_step_h4 in stm_sgm_step.h).
//
// A wave is four 16-lane DPP rows, each holding one pixel's 128 disparities, eight per lane.  As in
// sgm_tile_h2_kernel four waves own a tile and take it through the four direction pairs, a wave
// two lines of a pair -- but here the lines' forward chains live in rows 0, 1 and their inverse
// chains in rows 2, 3 of ONE instruction stream: step k puts the forward chains on slot k and the
// inverse chains on slot 7 - k.  What that buys is the cross-lane part of a step (minimum over a
// pixel's lanes, neighbour values, penalty broadcast): 7 instructions per four pixel-steps
// instead of 12 per two -- the kernel is bound by vector instructions issued.
//
// The update S = (S + F) + I of a slot needs F and I in the same lanes.  After step k >= 4 the
// registers L[k] = (F(k) | I(7 - k)) and L[7 - k] = (F(7 - k) | I(k)) are complete; one
// v_permlane32_swap per register turns them into (F(k) | F(7 - k)) and (I(7 - k) | I(k)), and two
// rounds through the sum tile follow, every lane busy in both: each lane adds its F to the sum of
// its OWN slot, then its I to the sum of its PARTNER's slot (written a moment ago by the partner
// lanes of the same wave: LDS executes a wave's accesses in order).
//
// Reference semantics: src/semiglobal.cpp:116-159 (recurrence), :191-194 (order of the passes).
#include "stm_sgm_tile_d128.h"

namespace stm {
namespace {

// Per-lane constants of a pair.  The lane's row r = lane / 16 names its line (r & 1: the wave's
// first or second line, as H2Pair's `half`) and its chain (r / 2: forward or inverse).
struct H4Pair {
  unsigned own[2];          // tile pixel (y * 8 + x) of the lane's chain at step k, one byte per step (its
                            // partner chain's -- same line, other direction -- are the lanes' 32 away)
  int e_init, e_rst;        // entry states at the start / where the line's second part begins: offset
                            // in floats | 1 when it lies in the row border states (H2Pair::eoff)
  bool restart;             // the second part begins at an entry state (it is another diagonal)
};

template <int PHASE>
__device__ __forceinline__ void h4_pair_init(H4Pair &K, const TiledGeom &G, int w, int lane) {
  constexpr int pstep = PhaseDir<PHASE>::dy * kT + PhaseDir<PHASE>::dx;
  const int sel = (lane >> 4) & 1;
  const bool inv = lane >= 32;
  H2Pair L;
  h2_pair_init<PHASE>(L, G, w, sel);
  // Slots of the line: [0, na) its first part from pix0, [na, 8) its second from pix1.
  const int na = h2_na(PHASE, w);
  K.own[0] = K.own[1] = 0u;
#pragma unroll
  for (int k = 0; k < kT; ++k) {
    const int sf = k, si = kT - 1 - k;  // slot of the forward / inverse chain at step k
    const int pf = sf < na ? L.pix0 + sf * pstep : L.pix1 + (sf - na) * pstep;
    const int pi = si < na ? L.pix0 + si * pstep : L.pix1 + (si - na) * pstep;
    K.own[k >> 2] |= (unsigned)(inv ? pi : pf) << (8 * (k & 3));
  }
  // Forward rows enter at the first part's F entry and restart at the second part's; inverse rows
  // enter at the second part's I entry (the first part's where the line is one piece) and restart
  // at the first part's.
  K.e_init = inv ? (na < kT ? L.eoff[3] : L.eoff[1]) : L.eoff[0];
  K.e_rst = inv ? L.eoff[1] : L.eoff[2];
  K.restart = L.restart;
}

__device__ __forceinline__ Row<8> h4_load_entry(int eoff, const H2TileBase &T, int l16) {
  // (the offsets are relative to an interior tile's terms: added to anything else they may leave
  // the workspace)
  const float *p = T.real ? ((eoff & 1) ? T.row : T.col) + (eoff & ~1) : T.row;
  return load_row<8, true>(p, 0, 128, l16 * 8);
}

// Penalties of the lane's chain: lane i of its row holds the one of step i & 7 -- the step into
// the chain's slot of that step from its predecessor on the chain.
template <int PHASE>
__device__ __forceinline__ float h4_penalties(const H4Pair &K, const float *Ip, float p1, float p2, int lane) {
  constexpr int dx = PhaseDir<PHASE>::dx, dy = PhaseDir<PHASE>::dy;
  const int s = lane & 7;
  const bool inv = lane >= 32;
  const int pix = (int)((K.own[s >> 2] >> (8 * (s & 3))) & 0xffu);
  const int px = pix & (kT - 1), py = pix >> 3;
  const int qx = px + (inv ? dx : -dx), qy = py + (inv ? dy : -dy);
  return sgm_penalty(Ip[(py + 1) * (kT + 2) + px + 1], Ip[(qy + 1) * (kT + 2) + qx + 1], p1, p2);
}

__device__ __forceinline__ Row<8> h4_lds_row(const float *tile, int byte_addr) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f a = *reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(tile) + byte_addr);
  const v4f b = *reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(tile) + byte_addr + 16);
  Row<8> r;
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  return r;
}
__device__ __forceinline__ void h4_lds_store(float *tile, int byte_addr, const Row<8> &r) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f a, b;
  a.x = r.v[0]; a.y = r.v[1]; a.z = r.v[2]; a.w = r.v[3];
  b.x = r.v[4]; b.y = r.v[5]; b.z = r.v[6]; b.w = r.v[7];
  *reinterpret_cast<v4f *>(reinterpret_cast<char *>(tile) + byte_addr) = a;
  *reinterpret_cast<v4f *>(reinterpret_cast<char *>(tile) + byte_addr + 16) = b;
}

// x's lanes 32-63 <-> y's lanes 0-31, register by register.
__device__ __forceinline__ void h4_swap_halves(Row<8> &x, Row<8> &y) {
  asm("s_nop 1\n\t"
      "v_permlane32_swap_b32 %0, %8\n\t"
      "v_permlane32_swap_b32 %1, %9\n\t"
      "v_permlane32_swap_b32 %2, %10\n\t"
      "v_permlane32_swap_b32 %3, %11\n\t"
      "v_permlane32_swap_b32 %4, %12\n\t"
      "v_permlane32_swap_b32 %5, %13\n\t"
      "v_permlane32_swap_b32 %6, %14\n\t"
      "v_permlane32_swap_b32 %7, %15"
      : "+v"(x.v[0]), "+v"(x.v[1]), "+v"(x.v[2]), "+v"(x.v[3]), "+v"(x.v[4]), "+v"(x.v[5]), "+v"(x.v[6]), "+v"(x.v[7]),
        "+v"(y.v[0]), "+v"(y.v[1]), "+v"(y.v[2]), "+v"(y.v[3]), "+v"(y.v[4]), "+v"(y.v[5]), "+v"(y.v[6]), "+v"(y.v[7]));
}

// One pair of an interior tile for one wave: the forward and inverse chains of its two lines,
// `between` (the barrier behind which the previous pair's update is complete) in the middle of
// the chains, and from then on the update of two slots per step.
template <int PHASE, int NA, int FINAL, typename Between>
__device__ __forceinline__ void h4_pair(const H4Pair &K, const Row<8> &e_init, const Row<8> &e_rst, int cls, int w,
                                        const TiledGeom &G, int true_paths, int accumulate, const float *Ct, float *St,
                                        float pen, StepCtxH4 &X, int lane, int x0, int y0, float *__restrict__ out,
                                        Row<8> &exit_f, Between between) {
  using std::integral_constant;
  constexpr int kPixBytes = 128 * 4, kSumPixBytes = st_stride(128) * 4;
  const int l16 = lane & 15, lo32 = l16 * 32;
  const bool inv_row = lane >= 32;
  // cls (h2_count_class): 0 no inverse pass at all, 1 all of them once, 2 per line and part, 3 the
  // tile holds a diagonal that counts several times or before its forward pass.
  const bool run_inv = cls != 0;
  const bool counted = PHASE >= 2 && cls == 3;
  int pre0 = 0, pre1 = 0, inv0 = 1, inv1 = 1;  // seg_counts of the two parts of the lane's line (cls >= 2)
  if (PHASE >= 2 && cls >= 2) {
    H2Part q0, q1;
    bool restart_;
    h2_parts(PHASE, w, (lane >> 4) & 1, q0, q1, restart_);
    Seg s0, s1;
    s0.ex = q0.ex; s0.ey = q0.ey; s0.n = q0.n;
    s1.ex = q1.ex; s1.ey = q1.ey; s1.n = q1.n;
    const Counts c0 = seg_counts(PHASE, s0, x0, y0, G, true_paths), c1 = seg_counts(PHASE, s1, x0, y0, G, true_paths);
    pre0 = c0.pre; inv0 = c0.inv;
    pre1 = c1.pre; inv1 = c1.inv;
  }
  // (the tables are loop-invariant: left alone, hipcc forms the LDS addresses of every step of
  // every pair once per kernel and keeps them -- in scratch memory; the empty asm ties each
  // extraction to its use)
  auto own_pix = [&](auto k_) {
    constexpr int k = decltype(k_)::value;
    unsigned t = K.own[k >> 2];
    asm volatile("" : "+v"(t));
    return (int)((t >> (8 * (k & 3))) & 0xffu);
  };
  // The partner chain's pixels: the own table of the lanes 32 away (both halves exchanged).
  unsigned oth[2];
  {
    unsigned a0 = K.own[0], a1 = K.own[1], b0 = K.own[0], b1 = K.own[1];
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3"
        : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
    // a = (own.lo | own.lo), b = (own.hi | own.hi)
    oth[0] = inv_row ? a0 : b0;
    oth[1] = inv_row ? a1 : b1;
  }
  auto oth_pix = [&](auto k_) {
    constexpr int k = decltype(k_)::value;
    unsigned t = oth[k >> 2];
    asm volatile("" : "+v"(t));
    return (int)((t >> (8 * (k & 3))) & 0xffu);
  };
  Row<8> L[kT];
  Row<8> st = e_init;
  // The update of the slots k and 7 - k (k >= 4): L[k] = (F(k) | I(7 - k)), L[7 - k] = (F(7 - k) | I(k)).
  auto update = [&](auto k_) {
    constexpr int k = decltype(k_)::value, r = kT - 1 - k;
    Row<8> &xf = L[k], &yi = L[r];
    h4_swap_halves(xf, yi);  // xf = (F(k) | F(7 - k)): every lane's OWN slot; yi = (I(7 - k) | I(k)): its PARTNER's
    // The part of the line the lane's own / partner slot lies in (own: k for forward rows, 7 - k
    // for inverse rows; the partner's the other way round).
    const int sp = inv_row ? k : r;
    const int own = own_pix(k_), othp = oth_pix(k_);
    const int a_own = own * kSumPixBytes + lo32, a_oth = othp * kSumPixBytes + lo32;
    if (counted) {
      // ((S + [I x pre]) + F) + I x inv with the line's own counts: the repeats before the
      // forward pass are the partner lanes' (they hold the slot's I) -- a round of its own.
      const int pre = sp < NA ? pre0 : pre1;
      int npre = __builtin_amdgcn_readlane(pre, 0);  // wave-uniform loop bound: the largest of the four rows
      npre = max(npre, __builtin_amdgcn_readlane(pre, 16));
      npre = max(npre, __builtin_amdgcn_readlane(pre, 32));
      npre = max(npre, __builtin_amdgcn_readlane(pre, 48));
      if (npre > 0) {
        Row<8> acc = h4_lds_row(St, a_oth);
        for (int i = 0; i < npre; ++i) {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc.v[j] = i < pre ? acc.v[j] + yi.v[j] : acc.v[j];
        }
        h4_lds_store(St, a_oth, acc);
        wave_lds_handoff();
      }
    }
    {  // round F: the lane's own slot
      Row<8> acc;
      if (PHASE == 0 && !accumulate) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc.v[j] = 0.0f;
      } else {
        acc = h4_lds_row(St, a_own);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc.v[j] = acc.v[j] + xf.v[j];
      // (without inverse passes this is the pair's result)
      if (PHASE < 3 || run_inv || FINAL != 0) h4_lds_store(St, a_own, acc);
      if (PHASE == 3 && FINAL != 2 && !run_inv)
        store_row<8, true>(out, (size_t)tile_out_pixel(G, x0, y0, own), 128, l16 * 8, acc);
    }
    if (run_inv) {  // round I: the partner's slot, behind the partner lanes' round F
      wave_lds_handoff();
      Row<8> acc = h4_lds_row(St, a_oth);
      if (PHASE < 2 || cls == 1) {  // (wave-uniform) every inverse pass counts once
#pragma unroll
        for (int j = 0; j < 8; ++j) acc.v[j] = acc.v[j] + yi.v[j];
      } else if (!counted) {        // some lines' inverse passes do not count at all
        const bool run = (sp < NA ? inv0 : inv1) != 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = acc.v[j] + yi.v[j];
          acc.v[j] = run ? v : acc.v[j];
        }
      } else {
        const int inv = sp < NA ? inv0 : inv1;
        int ninv = __builtin_amdgcn_readlane(inv, 0);
        ninv = max(ninv, __builtin_amdgcn_readlane(inv, 16));
        ninv = max(ninv, __builtin_amdgcn_readlane(inv, 32));
        ninv = max(ninv, __builtin_amdgcn_readlane(inv, 48));
        for (int i = 0; i < ninv; ++i) {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc.v[j] = i < inv ? acc.v[j] + yi.v[j] : acc.v[j];
        }
      }
      if (PHASE < 3 || FINAL != 0) h4_lds_store(St, a_oth, acc);
      if (PHASE == 3 && FINAL != 2) store_row<8, true>(out, (size_t)tile_out_pixel(G, x0, y0, othp), 128, l16 * 8, acc);
    }
  };
  Row<8> cn = h4_lds_row(Ct, own_pix(integral_constant<int, 0>()) * kPixBytes + lo32);
  auto step = [&](auto k_) {
    constexpr int k = decltype(k_)::value;
    // The line's second part begins: forward rows at slot NA, inverse rows one step after slot NA.
    if constexpr (PHASE >= 2 && k > 0 && (k == NA || k == kT - NA)) {
      const bool here = K.restart && (inv_row ? k == kT - NA : k == NA);
      st = pick_row(here, e_rst, st);
    }
    const Row<8> c = cn;
    if constexpr (k < kT - 1) cn = h4_lds_row(Ct, own_pix(integral_constant<int, k + 1>()) * kPixBytes + lo32);
    if constexpr (k == kT / 2) between();
    sgm_step_h4<k>(st, c, pen, X);
    L[k] = st;
    if constexpr (PHASE == 0 && k == kT - 1) exit_f = st;  // rows 0, 1: the rightward state where the line leaves the tile
    if constexpr (k >= kT / 2) update(k_);
  };
  step(integral_constant<int, 0>()); step(integral_constant<int, 1>());
  step(integral_constant<int, 2>()); step(integral_constant<int, 3>());
  step(integral_constant<int, 4>()); step(integral_constant<int, 5>());
  step(integral_constant<int, 6>()); step(integral_constant<int, 7>());
}

// The pair with the wave's slot count as a constant (the diagonal pairs change part at slot
// w + 1, wave 3 at slot 4).
template <int PHASE, int FINAL, typename Between>
__device__ __forceinline__ void h4_pair_w(int w, const H4Pair &K, const Row<8> &e_init, const Row<8> &e_rst, int cls,
                                          const TileArgs &P, const float *Ct, float *St, float pen, StepCtxH4 &X,
                                          int lane, int x0, int y0, float *__restrict__ out, Row<8> &exit_f,
                                          Between between) {
#define STM_H4_PAIR(NA) \
  h4_pair<PHASE, NA, FINAL>(K, e_init, e_rst, cls, w, P.G, P.true_paths, P.accumulate, Ct, St, pen, X, lane, x0, y0, out, exit_f, between)
  if constexpr (PHASE < 2) {
    STM_H4_PAIR(kT);
  } else {
    switch (w) {
      case 0: STM_H4_PAIR(1); break;
      case 1: STM_H4_PAIR(2); break;
      case 2: STM_H4_PAIR(3); break;
      default: STM_H4_PAIR(4); break;
    }
  }
#undef STM_H4_PAIR
}

template <int FINAL>
__global__ __launch_bounds__(4 * 64, 2) void sgm_tile_h4_kernel(const float *__restrict__ cost_all,
                                                                 const float *__restrict__ image_all,
                                                                 float *__restrict__ out_all,
                                                                 int32_t *__restrict__ disp_all,
                                                                 float *__restrict__ ws, const TileArgs P) {
  constexpr int D = 128;
  __shared__ __attribute__((aligned(16))) float Ct[kT * kT * D];
  __shared__ __attribute__((aligned(16))) float St[kT * kT * st_stride(D)];
  __shared__ float Ip[(kT + 2) * (kT + 2)];
  const TiledGeom &G = P.G;
  const int lane = threadIdx.x & 63;
  const int w = wave_uniform((int)(threadIdx.x >> 6));
  const int half = lane >> 5, hl = lane & 31;  // staging: tile row 2 w + half, four values per lane
  const int d0 = hl * 4, hl16 = hl * 16;
  const int l16 = lane & 15;
  const bool f_row = lane < 32;
  const size_t npix = (size_t)G.H * G.W;
  const int stride = (int)gridDim.x;
  StepCtxH4 X;
  X.init(P.p1, l16, P.dlim);
  StepCtx X1;  // the generic forms: one pixel per wave, two values per lane
  X1.init(P.p1, lane * 2, P.dlim, lane == 63 && P.dlim < D);
  H4Pair K0, K1, K2, K3;
  h4_pair_init<0>(K0, G, w, lane);
  h4_pair_init<1>(K1, G, w, lane);
  h4_pair_init<2>(K2, G, w, lane);
  h4_pair_init<3>(K3, G, w, lane);
  // Workgroups own whole tile rows and carry the rightward state along them behind row sweeps
  // (P.by_rows), single tiles behind border walks: as sgm_tile_h2_kernel.
  const bool by_rows = P.by_rows != 0;
  const int nunits = TileUnitWalk::units(G, by_rows);
  TileUnitWalk cur_t, next_t;  // the tile being computed / the one being prefetched
  cur_t.init((int)blockIdx.x, G, by_rows);
  if (cur_t.unit >= nunits) return;
  next_t = cur_t;

  auto interior = [&](const TileUnitWalk &t) { return tile_interior(G, t.trow, t.tcol); };
  Row<4> cpre[kT];  // next tile: tile row 2 w + half of the cost tile
  float ipre = 0.0f;
  Row<8> E0, E1;    // next tile: entry states of pairs 0 and 1 (interior tiles)
  Row<8> carry;     // rows 0, 1: the previous tile's rightward exit states of the tile rows 2 w, 2 w + 1
  bool have_carry = false;
  // Every prefetch redefines all of its registers unconditionally (past the last tile it
  // re-reads that tile; rows and samples outside a partial tile are clamped into it; the
  // entry states of a tile that is not interior are read at the workspace's first border row).
  auto prefetch_cost = [&](const TileUnitWalk &t) {
    const int x0 = t.tcol * kT, y0 = t.trow * kT;
    const int tw = imin(kT, G.W - x0), th = imin(kT, G.H - y0);
    const float *cost = cost_all + (size_t)t.b * npix * D;
#pragma unroll
    for (int c = 0; c < kT; ++c)
      cpre[c] = load_row_at<4, true>(cost, (long)(tile_cost_pixel(G, x0, y0, tw, th, 2 * w + half, c) * (unsigned)D), D, d0);
    ipre = image_all[(size_t)t.b * npix +
                     tile_intensity_pixel(G, x0, y0, imin((int)threadIdx.x, (kT + 2) * (kT + 2) - 1))];
  };
  auto entry_base = [&](const TileUnitWalk &t) {
    H2TileBase T = h2_tile_base(ws, G, t);
    if (!interior(t)) {  // a valid dummy address: the first border row of the workspace
      T.row = ws + G.rowb_off;
      T.col = T.row;
      T.real = false;
    }
    return T;
  };
  prefetch_cost(next_t);
  E0 = h4_load_entry(K0.e_init, entry_base(next_t), l16);
  E1 = h4_load_entry(K1.e_init, entry_base(next_t), l16);
  for (;;) {
    const int b = cur_t.b, x0 = cur_t.tcol * kT, y0 = cur_t.trow * kT;
    const int tw = imin(kT, G.W - x0), th = imin(kT, G.H - y0);
    float *out = out_all ? out_all + (size_t)b * npix * D : nullptr;
    int32_t *disp = disp_all ? disp_all + (size_t)b * npix : nullptr;
    STM_STAMP_TILE_ID((cur_t.b * G.nTR + cur_t.trow) * G.nTC + cur_t.tcol);
    STM_STAMP_WAVE_ID(w);
    STM_STAMP(0);
    const int cls2 = h2_count_class(2, G, x0, y0, P.true_paths), cls3 = h2_count_class(3, G, x0, y0, P.true_paths);
    const bool fast = interior(cur_t);
    if (2 * w + half < th) {
#pragma unroll
      for (int c = 0; c < kT; ++c) h2_lds_store(Ct, ((2 * w + half) * kT + c) * D * 4 + hl16, cpre[c]);
    }
    if (threadIdx.x < (kT + 2) * (kT + 2)) Ip[threadIdx.x] = ipre;
    if (P.accumulate) {  // (STM_SGM_ACCUMULATE: as sgm_tile_h2_kernel)
      stage_accumulate_row<2>(St, out, G, x0, y0, tw, th, 2 * w, lane * 2);
      stage_accumulate_row<2>(St, out, G, x0, y0, tw, th, 2 * w + 1, lane * 2);
    }
    const Row<8> Ea = pick_row(have_carry && f_row, carry, E0);
    const Row<8> Eb = E1;
    STM_STAMP(1);
    lds_barrier();
    STM_STAMP(2);
    bool more;
    {
      TileUnitWalk n = cur_t;
      n.next(G, stride, by_rows);
      more = n.unit < nunits;
      if (more) next_t = n;
    }
    // Where the rows of this tile leave it to the right (column border states of direction 0).
    float *exits = colb_ptr(ws, G, b, 0, cur_t.tcol, y0);
    if (fast) {
      // The entry states of pairs 2 and 3 are requested now (a global load takes several times as
      // long as a pair); the next tile's cost rows and its pairs 0 and 1 once pairs 0 and 1 are done.
      // (results return in issue order and every set is eight registers: pair 2's entries now, its
      // restart entries and pair 3's entries behind pair 0, pair 3's restart entries behind pair 1)
      const H2TileBase T = h2_tile_base(ws, G, cur_t);
      const Row<8> Ec = h4_load_entry(K2.e_init, T, l16);
      // The penalties of all four pairs now: four independent division sequences.
      const float pen0 = h4_penalties<0>(K0, Ip, P.p1, P.p2, lane);
      const float pen1 = h4_penalties<1>(K1, Ip, P.p1, P.p2, lane);
      const float pen2 = h4_penalties<2>(K2, Ip, P.p1, P.p2, lane);
      const float pen3 = h4_penalties<3>(K3, Ip, P.p1, P.p2, lane);
      Row<8> unused;
      h4_pair_w<0, FINAL>(w, K0, Ea, Ea, 1, P, Ct, St, pen0, X, lane, x0, y0, out, carry, [] {});
      have_carry = by_rows;
      if (by_rows && f_row) store_row<8, true>(exits + (size_t)(2 * w + ((lane >> 4) & 1)) * D, 0, D, l16 * 8, carry);
      const Row<8> Ecr = h4_load_entry(K2.e_rst, T, l16);
      const Row<8> Ed = h4_load_entry(K3.e_init, T, l16);
      h4_pair_w<1, FINAL>(w, K1, Eb, Eb, 1, P, Ct, St, pen1, X, lane, x0, y0, out, unused, [] { lds_barrier(); });
      const Row<8> Edr = h4_load_entry(K3.e_rst, T, l16);
      h4_pair_w<2, FINAL>(w, K2, Ec, Ecr, cls2, P, Ct, St, pen2, X, lane, x0, y0, out, unused, [] { lds_barrier(); });
      // The next tile's cost rows and first entry states: 48 registers that pair 2, with its own
      // and pair 3's entry states live, does not have (a value spilled there is reloaded through
      // the vector memory pipe, in order behind every prefetch in flight).
      prefetch_cost(next_t);
      E0 = h4_load_entry(K0.e_init, entry_base(next_t), l16);
      E1 = h4_load_entry(K1.e_init, entry_base(next_t), l16);
      STM_STAMP(3);
      h4_pair_w<3, FINAL>(w, K3, Ed, Edr, cls3, P, Ct, St, pen3, X, lane, x0, y0, out, unused,
                          [&] { STM_STAMP(4); lds_barrier(); STM_STAMP(5); });
      STM_STAMP(6);
    } else {
      // (the previous tile of this row, if it took the other form, stored this tile's
      // rightward entry states a while ago: same wave, same addresses)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const BorderBase B = border_base(ws, G, b);
      have_carry = false;
      H2GenericEntries<0> Q0;
      H2GenericEntries<1> Q1;
      H2GenericEntries<2> Q2;
      H2GenericEntries<3> Q3;
      h2_generic_entries<0>(Q0, w, tw, th, P, B, x0, y0, lane);
      h2_generic_entries<1>(Q1, w, tw, th, P, B, x0, y0, lane);
      h2_generic_entries<2>(Q2, w, tw, th, P, B, x0, y0, lane);
      h2_generic_entries<3>(Q3, w, tw, th, P, B, x0, y0, lane);
      prefetch_cost(next_t);
      h2_pair_generic<0, FINAL>(Q0, w, tw, th, Ct, St, Ip, P, x0, y0, out, disp, X1, lane, by_rows ? exits : nullptr, [] {});
      h2_pair_generic<1, FINAL>(Q1, w, tw, th, Ct, St, Ip, P, x0, y0, out, disp, X1, lane, nullptr,
                                [] { lds_barrier(); });
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ... and this tile's for the next one
      E0 = h4_load_entry(K0.e_init, entry_base(next_t), l16);
      E1 = h4_load_entry(K1.e_init, entry_base(next_t), l16);
      h2_pair_generic<2, FINAL>(Q2, w, tw, th, Ct, St, Ip, P, x0, y0, out, disp, X1, lane, nullptr,
                                [] { lds_barrier(); });
      h2_pair_generic<3, FINAL>(Q3, w, tw, th, Ct, St, Ip, P, x0, y0, out, disp, X1, lane, nullptr,
                                [] { lds_barrier(); });
    }
    lds_barrier();
    STM_STAMP(7);
    // Winner takes all of the finished sum tile (sixteen pixels per wave); the next tile's
    // first update of the sum tile lies behind its staging barrier.
    if constexpr (FINAL != 0) tile_wta_lds<D, 4>(St, w, lane, tw, th, G, x0, y0, disp);
    if (!more) break;
    cur_t = next_t;
  }
}

}  // namespace

int tiled_launch_tiles_h4(const float *cost, const float *left, float *out, int32_t *disp, float *ws,
                          const TiledCall &c, int final_mode, hipStream_t s) {
  const TiledGeom &G = c.G;
  TileArgs A;
  A.G = G; A.p1 = c.p1; A.p2 = c.p2; A.dlim = c.dlim; A.true_paths = c.true_paths; A.accumulate = c.accumulate;
  A.stamps = c.tile_stamps;
  A.by_rows = c.by_rows ? 1 : 0;
  // Four-wave workgroups, two per CU, each owning whole tile rows (or single tiles).
  const long long wgs = (long long)c.cus * 2, units = TileUnitWalk::units(G, c.by_rows);
  const dim3 grid((unsigned)(units < wgs ? units : wgs)), block(4 * 64);
  if (final_mode == 0)
    hipLaunchKernelGGL((sgm_tile_h4_kernel<0>), grid, block, 0, s, cost, left, out, disp, ws, A);
  else if (final_mode == 1)
    hipLaunchKernelGGL((sgm_tile_h4_kernel<1>), grid, block, 0, s, cost, left, out, disp, ws, A);
  else
    hipLaunchKernelGGL((sgm_tile_h4_kernel<2>), grid, block, 0, s, cost, left, out, disp, ws, A);
  return check_launch("stm_sgm_f32 (tiled: tiles)");
}

}  // namespace stm
