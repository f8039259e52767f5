// Semiglobal aggregation, tiled form: the tile kernel for D = 128, two lines of a direction
// pair per wave (see stm_sgm_tiled.hip for the whole form; sgm_step_h2* in stm_sgm_step.h).
#include "stm_sgm_tile_d128.h"

namespace stm {
namespace {

template <int N>
__device__ __forceinline__ void h2_load_entries(Row<4> (&E)[4], const H2Pair &K, const H2TileBase &T, int d0) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    // (the offsets are relative to an interior tile's terms: added to anything else they may
    // leave the workspace -- e.g. by -(H + 1) rows of D for a tall image)
    const float *p = T.real ? ((K.eoff[i] & 1) ? T.row : T.col) + (K.eoff[i] & ~1) : T.row;
    E[i] = load_row<4, true>(p, 0, 128, d0);
  }
}

// The lines' constants (H2Pair: the same for the 32 lanes of a half-wave) live in LDS, one row of
// eight words per (pair, half-wave), and are fetched where they are used: the kernel runs at the
// 256-register limit, and what hipcc spills around the tile loop it reloads from scratch memory
// with an s_waitcnt vmcnt(0) -- behind the next tile's cost rows and entry states, requested
// ahead on purpose, i.e. a memory round trip per reload (as in the skewed-tile form's forward
// pass, stm_sgm_skew_body.h).  An LDS read waits for LDS reads only.
constexpr int kKtN = 8;
template <int PHASE>
__device__ __forceinline__ void h2_kt_fill(int *Kt, const TiledGeom &G, int w, int half, int hl) {
  H2Pair K;
  h2_pair_init<PHASE>(K, G, w, half);
  if (hl == 0) {
    int *row = Kt + ((PHASE * 4 + w) * 2 + half) * kKtN;
    row[0] = K.pix0; row[1] = K.pix1; row[2] = K.restart ? 1 : 0; row[3] = 0;
    row[4] = K.eoff[0]; row[5] = K.eoff[1]; row[6] = K.eoff[2]; row[7] = K.eoff[3];
  }
}
// (every fetch is a new one: nothing of an earlier fetch stays in registers for it; what the
// caller does not use is not loaded)
template <int PHASE>
__device__ __forceinline__ H2Pair h2_kt_fetch(const int *Kt, int w, int half) {
  int o = ((PHASE * 4 + w) * 2 + half) * kKtN;
  asm volatile("" : "+v"(o));
  const int *row = Kt + o;
  H2Pair K;
  K.pix0 = row[0]; K.pix1 = row[1]; K.restart = row[2] != 0;
  K.eoff[0] = row[4]; K.eoff[1] = row[5]; K.eoff[2] = row[6]; K.eoff[3] = row[7];
  return K;
}

// Penalties of the lane's line: lane i of every 16-lane row holds the one of slot i & 7 on
// the F chain (i < 8: the step from its predecessor) or the I chain (from its successor).
template <int PHASE>
__device__ __forceinline__ float h2_penalties(const H2Pair &K, int NA, const float *Ip, float p1, float p2, int lane) {
  constexpr int dx = PhaseDir<PHASE>::dx, dy = PhaseDir<PHASE>::dy;
  constexpr int pstep = dy * kT + dx;
  const int s = lane & 7;
  const bool inv = (lane >> 3) & 1;
  const int pix = s < NA ? K.pix0 + s * pstep : K.pix1 + (s - NA) * pstep;
  const int px = pix & (kT - 1), py = pix >> 3;
  const int qx = px + (inv ? dx : -dx), qy = py + (inv ? dy : -dy);
  return sgm_penalty(Ip[(py + 1) * (kT + 2) + px + 1], Ip[(qy + 1) * (kT + 2) + qx + 1], p1, p2);
}

// One pair of an interior tile for one wave (two lines): both chains, `between` (the barrier
// behind which the previous pair's update is complete), the update S = (S + F) + I.
template <int PHASE, int NA, int FINAL, typename Between>
__device__ __forceinline__ void h2_pair(const H2Pair &K, const Row<4> (&E)[4], int cls, int w, int half, const TiledGeom &G,
                                        int true_paths, int accumulate, const float *Ct,
                                        float *St, float pen, StepCtxH2 &X, int lane,
                                        int hl16, int W, int x0, int y0, float *__restrict__ out,
                                        int32_t *__restrict__ disp, Row<4> &exit_f, Between between) {
  constexpr int dx = PhaseDir<PHASE>::dx, dy = PhaseDir<PHASE>::dy;
  constexpr int pstep = dy * kT + dx;
  constexpr int kPixBytes = 128 * 4;
  StepCtxH2 XI = X;  // the inverse chain's cross-lane scratch (its own +inf edge lanes)
  // cls (h2_count_class): 0 no inverse pass at all, 1 all of them, 2 per line and part
  const bool run_inv = cls != 0;
  bool run0 = true, run1 = true;
  int pre0 = 0, pre1 = 0, inv0 = 1, inv1 = 1;  // seg_counts of the lane's two parts (cls >= 2)
  if (PHASE >= 2 && cls >= 2) {
    H2Part q0, q1;
    bool restart_;
    h2_parts(PHASE, w, half, q0, q1, restart_);
    Seg s0, s1;
    s0.ex = q0.ex; s0.ey = q0.ey; s0.n = q0.n;
    s1.ex = q1.ex; s1.ey = q1.ey; s1.n = q1.n;
    const Counts c0 = seg_counts(PHASE, s0, x0, y0, G, true_paths), c1 = seg_counts(PHASE, s1, x0, y0, G, true_paths);
    pre0 = c0.pre; inv0 = c0.inv;
    pre1 = c1.pre; inv1 = c1.inv;
    run0 = inv0 != 0;
    run1 = inv1 != 0;
  }
  const bool counted = PHASE >= 2 && cls == 3;
  const int a0 = K.pix0 * kPixBytes + hl16, a1 = K.pix1 * kPixBytes + hl16;
  auto addr = [&](auto s_) {  // LDS byte address of slot s in the cost tile (the slot part is an immediate)
    constexpr int s = decltype(s_)::value;
    return s < NA ? a0 + s * pstep * kPixBytes : a1 + (s - NA) * pstep * kPixBytes;
  };
  constexpr int kSumPixBytes = st_stride(128) * 4;  // the sum tile's pixels are padded (st_stride)
  const int sa0 = K.pix0 * kSumPixBytes + hl16, sa1 = K.pix1 * kSumPixBytes + hl16;
  auto saddr = [&](auto s_) {  // ... and in the sum tile
    constexpr int s = decltype(s_)::value;
    return s < NA ? sa0 + s * pstep * kSumPixBytes : sa1 + (s - NA) * pstep * kSumPixBytes;
  };
  Row<4> LF[kT], LI[kT];
  Row<4> pf = E[0];
  Row<4> pi = NA < kT ? E[3] : E[1];
  using std::integral_constant;
  // The update of one slot, S = (S + F) + I; `acc` is the slot's sum as read from the sum tile
  // (pair 0: nothing to read).  Pair 3 leaves the result in `acc` for `finish2`.
  auto update = [&](auto s_, Row<4> &acc) {
    constexpr int s = decltype(s_)::value;
    if (PHASE == 0 && !accumulate) {  // (accumulating: `acc` is the staged `out`, read like any other pair's sum)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc.v[j] = 0.0f;
    }
    f32x2 lo = f32x2{acc.v[0], acc.v[1]}, hi = f32x2{acc.v[2], acc.v[3]};
    const f32x2 ilo = f32x2{LI[s].v[0], LI[s].v[1]}, ihi = f32x2{LI[s].v[2], LI[s].v[3]};
    if (counted) {
      // S = ((S + [I x pre]) + F) + I x inv with the line's own counts (tile_add); the loop
      // bounds are the larger of the two halves' (a half holds one line).
      const int pre = s < NA ? pre0 : pre1, inv = s < NA ? inv0 : inv1;
      const int pa = __builtin_amdgcn_readlane(pre, 0), pb = __builtin_amdgcn_readlane(pre, 32);
      const int ia = __builtin_amdgcn_readlane(inv, 0), ib = __builtin_amdgcn_readlane(inv, 32);
      for (int r = 0, n = pa > pb ? pa : pb; r < n; ++r) {
        const f32x2 l2 = lo + ilo, h2 = hi + ihi;
        lo = r < pre ? l2 : lo;
        hi = r < pre ? h2 : hi;
      }
      lo = lo + f32x2{LF[s].v[0], LF[s].v[1]};
      hi = hi + f32x2{LF[s].v[2], LF[s].v[3]};
      for (int r = 0, n = ia > ib ? ia : ib; r < n; ++r) {
        const f32x2 l2 = lo + ilo, h2 = hi + ihi;
        lo = r < inv ? l2 : lo;
        hi = r < inv ? h2 : hi;
      }
    } else {
      lo = lo + f32x2{LF[s].v[0], LF[s].v[1]};
      hi = hi + f32x2{LF[s].v[2], LF[s].v[3]};
      if (run_inv) {
        const f32x2 lo2 = lo + ilo, hi2 = hi + ihi;
        const bool r = PHASE < 2 || (s < NA ? run0 : run1);
        lo = r ? lo2 : lo;
        hi = r ? hi2 : hi;
      }
    }
    acc.v[0] = lo.x; acc.v[1] = lo.y; acc.v[2] = hi.x; acc.v[3] = hi.y;
    // (pair 3: the finished sum stays in the sum tile for the winner-takes-all behind the tile's
    // last barrier, tile_wta_lds)
    if (PHASE < 3 || FINAL != 0) h2_lds_store(St, saddr(s_), acc);
  };
  // Pair 3: the finished sums of two slots leave (unless only their winners are wanted).
  auto finish2 = [&](auto sa_, const Row<4> &ra, auto sb_, const Row<4> &rb) {
    if constexpr (PHASE == 3 && FINAL != 2) {
      constexpr int sa = decltype(sa_)::value, sb = decltype(sb_)::value;
      const int d0 = hl16 >> 2;
      const int pa = sa < NA ? K.pix0 + sa * pstep : K.pix1 + (sa - NA) * pstep;
      const int pb = sb < NA ? K.pix0 + sb * pstep : K.pix1 + (sb - NA) * pstep;
      const unsigned qa = tile_out_pixel(G, x0, y0, pa), qb = tile_out_pixel(G, x0, y0, pb);  // < H * W
      store_row<4, true>(out, (size_t)qa, 128, d0, ra);
      store_row<4, true>(out, (size_t)qb, 128, d0, rb);
    }
  };
  auto sum_of = [&](auto s_) {
    Row<4> r;
    if (PHASE != 0 || accumulate) r = h2_lds_row(St, saddr(s_));
    else r = E[0];  // (unused: any defined value)
    return r;
  };
  if (run_inv) {
    // Slot s of the forward chain together with slot 7 - s of the inverse chain
    // (sgm_step_h2x2); the cost rows of the next two slots are requested first.  From the
    // fifth step on both results of two slots exist (s and 7 - s): their update runs between
    // the steps, its LDS round trip under the next step's arithmetic -- after `between`, which
    // therefore sits in the middle of the chains.
    Row<4> cf = h2_lds_row(Ct, addr(integral_constant<int, 0>()));
    Row<4> ci = h2_lds_row(Ct, addr(integral_constant<int, kT - 1>()));
    Row<4> sa, sb;  // the sums of the two slots updated next
    auto both = [&](auto s_) {
      constexpr int s = decltype(s_)::value, r = kT - 1 - s;
      if constexpr (PHASE >= 2 && s == NA && s > 0) {
        if (K.restart) pf = E[2];
      }
      if constexpr (PHASE >= 2 && r == NA - 1 && r < kT - 1) {
        if (K.restart) pi = E[1];
      }
      Row<4> cfn = cf, cin = ci;
      if constexpr (s < kT - 1) {
        cfn = h2_lds_row(Ct, addr(integral_constant<int, s + 1>()));
        cin = h2_lds_row(Ct, addr(integral_constant<int, r - 1>()));
      }
      if constexpr (s == kT / 2) {
        between();
        sa = sum_of(integral_constant<int, s>());
        sb = sum_of(integral_constant<int, r>());
      }
      sgm_step_h2x2<s, 8 + r>(pf, cf, pi, ci, pen, X, XI);
      LF[s] = pf;
      LI[r] = pi;
      cf = cfn;
      ci = cin;
      if constexpr (s >= kT / 2) {
        update(integral_constant<int, s>(), sa);
        update(integral_constant<int, r>(), sb);
        finish2(integral_constant<int, s>(), sa, integral_constant<int, r>(), sb);
        if constexpr (s < kT - 1) {
          sa = sum_of(integral_constant<int, s + 1>());
          sb = sum_of(integral_constant<int, r - 1>());
        }
      }
    };
    both(integral_constant<int, 0>()); both(integral_constant<int, 1>());
    both(integral_constant<int, 2>()); both(integral_constant<int, 3>());
    both(integral_constant<int, 4>()); both(integral_constant<int, 5>());
    both(integral_constant<int, 6>()); both(integral_constant<int, 7>());
    if (PHASE == 0) exit_f = LF[kT - 1];  // the rightward state where the lane's row leaves the tile
  } else {
    between();
    Row<4> sa = sum_of(integral_constant<int, 0>()), held = sa;
    auto fstep = [&](auto s_) {
      constexpr int s = decltype(s_)::value;
      if constexpr (PHASE >= 2 && s == NA && s > 0) {
        if (K.restart) pf = E[2];
      }
      const Row<4> c = h2_lds_row(Ct, addr(s_));
      pf = sgm_step_h2<s>(pf, c, pen, X);
      LF[s] = pf;
      if constexpr (PHASE == 3) {  // results leave two at a time
        if constexpr (s % 2 == 0) {
          update(s_, sa);
          held = sa;
        } else {
          update(s_, sa);
          finish2(integral_constant<int, s - 1>(), held, s_, sa);
        }
      } else {
        update(s_, sa);
      }
      if constexpr (s < kT - 1) sa = sum_of(integral_constant<int, s + 1>());
    };
    fstep(integral_constant<int, 0>()); fstep(integral_constant<int, 1>());
    fstep(integral_constant<int, 2>()); fstep(integral_constant<int, 3>());
    fstep(integral_constant<int, 4>()); fstep(integral_constant<int, 5>());
    fstep(integral_constant<int, 6>()); fstep(integral_constant<int, 7>());
    if (PHASE == 0) exit_f = LF[kT - 1];
  }
}

// The pair with the wave's slot count as a constant (the diagonal pairs change segment at
// slot w + 1, wave 3 at slot 4).
template <int PHASE, int FINAL, typename Between>
__device__ __forceinline__ void h2_pair_w(int w, int half, const H2Pair &K, const Row<4> (&E)[4], int cls,
                                          const TileArgs &P, const float *Ct, float *St, float pen,
                                          StepCtxH2 &X, int lane, int hl16, int x0, int y0,
                                          float *__restrict__ out, int32_t *__restrict__ disp, Row<4> &exit_f,
                                          Between between) {
#define STM_H2_PAIR(NA)                                                                              \
  h2_pair<PHASE, NA, FINAL>(K, E, cls, w, half, P.G, P.true_paths, P.accumulate, Ct, St, pen, X, lane, hl16, P.G.W, x0, y0, \
                            out, disp, exit_f, between)
  if constexpr (PHASE < 2) {
    STM_H2_PAIR(kT);
  } else {
    switch (w) {
      case 0: STM_H2_PAIR(1); break;
      case 1: STM_H2_PAIR(2); break;
      case 2: STM_H2_PAIR(3); break;
      default: STM_H2_PAIR(4); break;
    }
  }
#undef STM_H2_PAIR
}

template <int FINAL>
__global__ __launch_bounds__(4 * 64, 2) void sgm_tile_h2_kernel(const float *__restrict__ cost_all,
                                                                 const float *__restrict__ image_all,
                                                                 float *__restrict__ out_all,
                                                                 int32_t *__restrict__ disp_all,
                                                                 float *__restrict__ ws, const TileArgs P) {
  constexpr int D = 128;
  __shared__ __attribute__((aligned(16))) float Ct[kT * kT * D];
  __shared__ __attribute__((aligned(16))) float St[kT * kT * st_stride(D)];
  __shared__ float Ip[(kT + 2) * (kT + 2)];
  __shared__ __attribute__((aligned(16))) int Kt[4 * 4 * 2 * kKtN];
  const TiledGeom &G = P.G;
  const int lane = threadIdx.x & 63;
  const int w = wave_uniform((int)(threadIdx.x >> 6));
  const int half = lane >> 5, hl = lane & 31;
  const int d0 = hl * 4, hl16 = hl * 16;
  const size_t npix = (size_t)G.H * G.W;
  const int stride = (int)gridDim.x;
  StepCtxH2 X;
  X.init(P.p1, hl, P.dlim);
  StepCtx X1;  // the generic forms: one pixel per wave, two values per lane
  X1.init(P.p1, lane * 2, P.dlim, lane == 63 && P.dlim < D);
  h2_kt_fill<0>(Kt, G, w, half, hl);
  h2_kt_fill<1>(Kt, G, w, half, hl);
  h2_kt_fill<2>(Kt, G, w, half, hl);
  h2_kt_fill<3>(Kt, G, w, half, hl);
  __syncthreads();
#define K0 h2_kt_fetch<0>(Kt, w, half)
#define K1 h2_kt_fetch<1>(Kt, w, half)
#define K2 h2_kt_fetch<2>(Kt, w, half)
#define K3 h2_kt_fetch<3>(Kt, w, half)
  // A workgroup owns whole tile rows and walks them left to right: the rightward state with
  // which a row leaves a tile is the next tile's entry state and stays in registers (between
  // two tiles of this form; otherwise it goes through the column border states, which every
  // tile writes) -- there is no left-to-right horizontal walk at D = 128, i.e. one read of the
  // cost volume fewer.
  // ... when the call's border states come from row sweeps (P.by_rows); behind border walks
  // (single pairs, small batches) every direction's states are there and the workgroups take
  // single tiles, round robin: a pair has 47 tile rows for 512 workgroups.
  const bool by_rows = P.by_rows != 0;
  const int nunits = TileUnitWalk::units(G, by_rows);
  TileUnitWalk cur_t, next_t;  // the tile being computed / the one being prefetched
  cur_t.init((int)blockIdx.x, G, by_rows);
  if (cur_t.unit >= nunits) return;
  next_t = cur_t;

  // A tile is interior when all of its neighbours exist (every line enters from a border state).
  auto interior = [&](const TileUnitWalk &t) { return tile_interior(G, t.trow, t.tcol); };
  Row<4> cpre[kT];  // next tile: tile row 2 w + half of the cost tile
  float ipre = 0.0f;
  Row<4> E0[4], E1[4];  // next tile: entry states of pairs 0 and 1 (interior tiles)
  Row<4> carry;         // the previous tile's rightward exit state of row 2 w + half
  bool have_carry = false;
  // Every prefetch redefines all of its registers unconditionally (past the last tile it
  // re-reads that tile; rows and samples outside a partial tile are clamped into it; the
  // entry states of a tile that is not interior are read at the workspace's first border row).
  auto prefetch_cost = [&](const TileUnitWalk &t) {
    const int x0 = t.tcol * kT, y0 = t.trow * kT;
    const int tw = imin(kT, G.W - x0), th = imin(kT, G.H - y0);
    const float *cost = cost_all + (size_t)t.b * npix * D;
    // (rows and columns outside a partial tile are clamped into it: tile_cost_pixel)
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
  h2_load_entries<2>(E0, K0, entry_base(next_t), d0);
  h2_load_entries<2>(E1, K1, entry_base(next_t), d0);
  for (;;) {
    const int b = cur_t.b, x0 = cur_t.tcol * kT, y0 = cur_t.trow * kT;
    const int tw = imin(kT, G.W - x0), th = imin(kT, G.H - y0);
    float *out = out_all ? out_all + (size_t)b * npix * D : nullptr;
    int32_t *disp = disp_all ? disp_all + (size_t)b * npix : nullptr;
    STM_STAMP_TILE_ID((cur_t.b * G.nTR + cur_t.trow) * G.nTC + cur_t.tcol);
    STM_STAMP_WAVE_ID(w);
    STM_STAMP(0);
    const int cls2 = h2_count_class(2, G, x0, y0, P.true_paths), cls3 = h2_count_class(3, G, x0, y0, P.true_paths);
    const bool fast = interior(cur_t) && cls2 >= 0 && cls3 >= 0;
    if (2 * w + half < th) {
#pragma unroll
      for (int c = 0; c < kT; ++c) h2_lds_store(Ct, ((2 * w + half) * kT + c) * D * 4 + hl16, cpre[c]);
    }
    if (threadIdx.x < (kT + 2) * (kT + 2)) Ip[threadIdx.x] = ipre;
    // (STM_SGM_ACCUMULATE: the caller's `out` becomes the sum tile's start.  A wave stages the
    // tile rows 2 w and 2 w + 1 -- the sixteen pixels whose winner it took from the previous tile's
    // sum, the sum tile's last reader, earlier in its own program.)
    if (P.accumulate) {
      stage_accumulate_row<2>(St, out, G, x0, y0, tw, th, 2 * w, lane * 2);
      stage_accumulate_row<2>(St, out, G, x0, y0, tw, th, 2 * w + 1, lane * 2);
    }
    Row<4> Ea[4], Eb[4];
    Ea[0] = pick_row(have_carry, carry, E0[0]);
    Ea[1] = E0[1];
    Eb[0] = E1[0]; Eb[1] = E1[1];
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
    // A global load takes several times as long as a pair (2-4 us against ~1 us), and results
    // return in issue order: the entry states of pairs 2 and 3 are requested now; the next tile's
    // cost rows and its pairs 0 and 1 follow once pairs 0 and 1 are done.
    if (fast) {
      const H2TileBase T = h2_tile_base(ws, G, cur_t);
      Row<4> Ec[4], Ed[4], unused;
      h2_load_entries<4>(Ec, K2, T, d0);
      h2_load_entries<4>(Ed, K3, T, d0);
      // The penalties of all four pairs now: four independent division sequences instead of one
      // in front of every pair's chains.
      const int na = h2_na(2, w);
      const float pen0 = h2_penalties<0>(K0, kT, Ip, P.p1, P.p2, lane);
      const float pen1 = h2_penalties<1>(K1, kT, Ip, P.p1, P.p2, lane);
      const float pen2 = h2_penalties<2>(K2, na, Ip, P.p1, P.p2, lane);
      const float pen3 = h2_penalties<3>(K3, na, Ip, P.p1, P.p2, lane);
      h2_pair_w<0, FINAL>(w, half, K0, Ea, 1, P, Ct, St, pen0, X, lane, hl16, x0, y0, out, disp, carry, [] {});
      have_carry = by_rows;
      if (by_rows) store_row<4, true>(exits + (size_t)(2 * w + half) * D, 0, D, d0, carry);
      h2_pair_w<1, FINAL>(w, half, K1, Eb, 1, P, Ct, St, pen1, X, lane, hl16, x0, y0, out, disp, unused,
                          [] { lds_barrier(); });
      // (the next tile's cost rows: 32 registers that pairs 0 and 1, with the entry states of all
      // four pairs live, do not have; two pairs are still ahead to cover the loads.  Measured
      // against requesting them at the start of the tile: 42.1 / 42.6 against 42.3 / 43.4 ms per
      // 256 Teddy pairs, fused / unfused)
      prefetch_cost(next_t);
      h2_load_entries<2>(E0, K0, entry_base(next_t), d0);
      h2_load_entries<2>(E1, K1, entry_base(next_t), d0);
      h2_pair_w<2, FINAL>(w, half, K2, Ec, cls2, P, Ct, St, pen2, X, lane, hl16, x0, y0, out, disp, unused,
                          [] { lds_barrier(); });
      STM_STAMP(3);
      h2_pair_w<3, FINAL>(w, half, K3, Ed, cls3, P, Ct, St, pen3, X, lane, hl16, x0, y0, out, disp, unused,
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
      h2_load_entries<2>(E0, K0, entry_base(next_t), d0);
      h2_load_entries<2>(E1, K1, entry_base(next_t), d0);
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
#undef K0
#undef K1
#undef K2
#undef K3
}


}  // namespace

int tiled_launch_tiles_h2(const float *cost, const float *left, float *out, int32_t *disp, float *ws,
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
    hipLaunchKernelGGL((sgm_tile_h2_kernel<0>), grid, block, 0, s, cost, left, out, disp, ws, A);
  else if (final_mode == 1)
    hipLaunchKernelGGL((sgm_tile_h2_kernel<1>), grid, block, 0, s, cost, left, out, disp, ws, A);
  else
    hipLaunchKernelGGL((sgm_tile_h2_kernel<2>), grid, block, 0, s, cost, left, out, disp, ws, A);
  return check_launch("stm_sgm_f32 (tiled: tiles)");
}

}  // namespace stm

