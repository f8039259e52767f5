// Device code shared by the two tile kernels of the tiled form of the semiglobal aggregation
// (stm_sgm_tile.hip: D = 64 / 256, one line of a pair per wave; stm_sgm_tile_h2.hip: D = 128, two
// lines per wave, whose border tiles run the forms here): entry states, the chains of a pair,
// the update of the sum tile, the winner-takes-all of a finished sum tile.
//
// Reference semantics: src/semiglobal.cpp:116-159 (recurrence), :191-194 (order of the passes).
#pragma once

#include <type_traits>

#include "stm_sgm_tiled_dev.h"

namespace stm {
namespace {

// =====================================================================================
// Tiles.
//
// Directions: 0 right, 1 left, 2 down, 3 up, 4 down-right, 5 up-left, 6 down-left, 7 up-right;
// phase p handles the pair (2p, 2p+1), a direction F and its inverse I, in the reference's
// order.  In a phase wave w owns up to two segments of F-paths inside the tile: row w, column
// w, or the two diagonals w and w + 8 (of 15), which together have 8 pixels.  Their pixels
// fill 8 "slots" in F order; the F chain runs through the slots upwards and the I chain
// downwards at the same time (independent dependency chains), restarting from a border
// state, or from L = C at the image border, where a segment begins.  Then every slot's sum is
// updated once: S = ((S + [I x pre]) + F) + I x count, the bracket and count != 1 only on the
// diagonals the reference re-walks or skips (Q6).
// =====================================================================================
struct TileArgs {
  TiledGeom G;
  float p1, p2;
  int dlim;
  int true_paths;
  int accumulate;  // STM_SGM_ACCUMULATE: the sum starts from the caller's contents of `out`, not from 0
  int by_rows;     // workgroups own tile rows (D = 128) / row segments (D = 64, 256) and carry the rightward
                   // state along them; no walk of that direction has run
  int segs;        // D = 64, 256: segments per tile row (row_segs)
  long long status_off;  // the workspace header (status word, ticket counter)
  unsigned long long *stamps;  // diagnostic builds only (stm_sgm_tiled_diag.h), else null
};

// An image's two blocks of border states; offsets inside a block fit 32 bits (make_geom).
struct BorderBase {
  const float *rowb, *colb;
};
__device__ __forceinline__ BorderBase border_base(float *ws, const TiledGeom &G, int b) {
  BorderBase B;
  B.rowb = ws + G.rowb_off + (long long)b * G.rowb_img;
  B.colb = ws + G.colb_off + (long long)b * G.colb_img;
  return B;
}

template <int PHASE>
struct PhaseDir {
  static constexpr int dx = phase_dx(PHASE);
  static constexpr int dy = phase_dy(PHASE);
};

// The entry states of a wave's phase: [0] F of segment a, [1] I of a and, for the diagonal
// pairs, [2] F of b, [3] I of b.  Bit i of `start` set: the path begins at that pixel (L = C),
// nothing is loaded.
template <int VPL, int PHASE>
struct Entries {
  static constexpr int N = PHASE >= 2 ? 4 : 2;
  Row<VPL> e[N];
  unsigned start;
};

template <int VPL, int PHASE>
__device__ __forceinline__ void load_entries(Entries<VPL, PHASE> &E, const PhaseGeom &g,
                                             const BorderBase &B, const TiledGeom &G, int x0, int y0,
                                             int d0) {
  constexpr int D = 64 * VPL;
  E.start = 0;
#pragma unroll
  for (int i = 0; i < Entries<VPL, PHASE>::N / 2; ++i) {
    const Seg &s = i == 0 ? g.a : g.b;
    // A segment that does not exist or begins at the image border reads a valid dummy address
    // (tile_entry: the first float of the image's row border block): the load stays
    // unconditional (one definition of the registers), its value is not used.
    const EntryRef f = tile_entry(G, PHASE, s, x0, y0, false), iv = tile_entry(G, PHASE, s, x0, y0, true);
    E.e[2 * i] = load_row<VPL, true>((f.is_row ? B.rowb : B.colb) + f.off, 0, D, d0);
    E.e[2 * i + 1] = load_row<VPL, true>((iv.is_row ? B.rowb : B.colb) + iv.off, 0, D, d0);
    if (!f.inside && s.n > 0) E.start |= 1u << (2 * i);
    if (!iv.inside && s.n > 0) E.start |= 2u << (2 * i);
  }
}

template <int VPL>
__device__ __forceinline__ Row<VPL> lds_row(const float *tile, int pix, int d0) {
  return load_row<VPL, true>(tile + (size_t)pix * (64 * VPL), 0, 64 * VPL, d0);
}

// The sum tile keeps 64 bytes of padding behind every pixel (stride D + 16 floats): the chains'
// accesses (whole pixels, 16 bytes per lane) do not care, and the winner-takes-all below, whose
// lanes of one instruction spread over several pixels, then finds them in different banks.
constexpr int st_stride(int D) { return D + 16; }
template <int VPL>
__device__ __forceinline__ Row<VPL> lds_srow(const float *St, int pix, int d0) {
  return load_row<VPL, true>(St + (size_t)pix * st_stride(64 * VPL), 0, 64 * VPL, d0);
}

// Winner takes all (src/winners_take_all.cu:21-40) of a finished sum tile in LDS, by all NWAVES
// waves of the workgroup: a wave takes 64 / NWAVES pixels, LP = NWAVES lanes share a pixel, lane
// r of them scans the 16-byte chunks r, r + LP, ... in ascending order with the reference's
// strict '<' (its first minimum; a NaN never wins), then the lanes of a pixel agree on the
// smallest value and on the lowest index among the lanes holding it.  A NaN at d = 0 wins, as in
// the reference scan, which starts from it and never replaces it.  The reduction used to sit in
// the last pair's update, one pixel per (half-)wave: a wave-wide minimum and a second one for
// the index per pixel, a fifth of the D = 128 tile's time.
template <int D, int NWAVES>
__device__ __forceinline__ void tile_wta_lds(const float *St, int wave, int lane, int tw, int th,
                                             const TiledGeom &G, int x0, int y0, int32_t *__restrict__ disp) {
  static_assert(NWAVES == 4 || NWAVES == 16, "4 or 16 lanes per pixel");
  constexpr int PPW = 64 / NWAVES, LP = NWAVES, NJ = D / (LP * 4);
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int pix = wave * PPW + lane / LP, r = lane % LP;
  const float *p = St + pix * st_stride(D) + r * 4;
  float best = __builtin_inff();
  int idxk = 0;  // the winner's index minus r * 4
  bool nan0 = false;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const v4f q = *reinterpret_cast<const v4f *>(p + j * LP * 4);
    const float v[4] = {q.x, q.y, q.z, q.w};
    if (j == 0) nan0 = r == 0 && v[0] != v[0];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool lt = v[e] < best;
      best = lt ? v[e] : best;
      idxk = lt ? j * LP * 4 + e : idxk;
    }
  }
  float m = best;
  m = vmin_f32(m, dpp_f32<0xb1>(m, m));  // quad_perm [1,0,3,2]
  m = vmin_f32(m, dpp_f32<0x4e>(m, m));  // quad_perm [2,3,0,1]
  if (LP == 16) {
    m = vmin_f32(m, dpp_f32<0x141>(m, m));  // row_half_mirror
    m = vmin_f32(m, dpp_f32<0x140>(m, m));  // row_mirror
  }
  int k = best == m ? idxk + r * 4 : 0x7fffffff;
  k = nan0 ? 0 : k;
  int t;
  t = dpp_i32<0xb1>(k, k); k = t < k ? t : k;
  t = dpp_i32<0x4e>(k, k); k = t < k ? t : k;
  if (LP == 16) {
    t = dpp_i32<0x141>(k, k); k = t < k ? t : k;
    t = dpp_i32<0x140>(k, k); k = t < k ? t : k;
  }
  const int px = pix & (kT - 1), py = pix >> 3;
  if (r == 0 && px < tw && py < th) disp[tile_out_pixel(G, x0, y0, pix)] = k;
}

// STM_SGM_ACCUMULATE: tile row `r` of the caller's `out` goes into the sum tile before the tile's
// first update (which then starts from it instead of from 0).  A compatibility path (the
// reference's `+=` into a pre-filled tensor): its loads are not prefetched.
template <int VPL>
__device__ __forceinline__ void stage_accumulate_row(float *St, const float *__restrict__ out, const TiledGeom &G,
                                                     int x0, int y0, int tw, int th, int r, int d0) {
  constexpr int D = 64 * VPL;
  if (r >= th) return;
  for (int c = 0; c < tw; ++c) {
    const Row<VPL> v = load_row<VPL, true>(out, (size_t)tile_out_pixel(G, x0, y0, r * kT + c), D, d0);
    store_row<VPL, true>(St + (size_t)(r * kT + c) * st_stride(D), 0, D, d0, v);
  }
}

// The slots of a wave's phase: pixel index (y * 8 + x, tile coordinates) of every slot.
template <int PHASE>
struct Slots {
  int pix[kT];
  int nA, nT;
  __device__ __forceinline__ void init(const PhaseGeom &g) {
    constexpr int dx = PhaseDir<PHASE>::dx, dy = PhaseDir<PHASE>::dy;
    nA = g.a.n;
    nT = g.a.n + g.b.n;
#pragma unroll
    for (int s = 0; s < kT; ++s) {
      const bool in_a = s < nA;
      const int k = in_a ? s : s - nA;
      const int x = (in_a ? g.a.ex : g.b.ex) + k * dx, y = (in_a ? g.a.ey : g.b.ey) + k * dy;
      pix[s] = s < nT ? y * kT + x : 0;
    }
  }
};

// Both chains of a phase: LF[s] / LI[s] = L of the forward / inverse direction at slot s.
template <int VPL, int PHASE>
__device__ __forceinline__ void tile_chains(const Slots<PHASE> &sl, const Entries<VPL, PHASE> &E, const float *Ct,
                                            const float *Ip, float p1, float p2, StepCtx &X, int lane,
                                            Row<VPL> (&LF)[kT], Row<VPL> (&LI)[kT]) {
  constexpr int dx = PhaseDir<PHASE>::dx, dy = PhaseDir<PHASE>::dy;
  const int d0 = lane * VPL;
  const int nA = sl.nA, nT = sl.nT;
  if (nT == 0) return;
  // Penalties: lane s (0..7) the step into slot s on the F chain (from its predecessor),
  // lane 8 + s the step into slot s on the I chain (from its successor).
  float pen;
  {
    const int s = imin(lane & 7, nT - 1);
    const bool inv = (lane >> 3) & 1;
    int pix = sl.pix[0];
#pragma unroll
    for (int k = 1; k < kT; ++k) pix = s == k ? sl.pix[k] : pix;
    const int px = pix & (kT - 1), py = pix >> 3;
    const int qx = px + (inv ? dx : -dx), qy = py + (inv ? dy : -dy);
    pen = sgm_penalty(Ip[(py + 1) * (kT + 2) + px + 1], Ip[(qy + 1) * (kT + 2) + qx + 1], p1, p2);
  }
  Row<VPL> pf = E.e[0], pi = E.e[1];
#pragma unroll
  for (int it = 0; it < kT; ++it) {
    {  // F chain, slot it
      const int s = it;
      if (s < nT) {
        const Row<VPL> c = lds_row<VPL>(Ct, sl.pix[s], d0);
        bool start = false;
        if (s == 0) {
          start = (E.start & 1u) != 0;
        } else if (PHASE >= 2 && s == nA) {  // the second diagonal begins
          if constexpr (PHASE >= 2) pf = E.e[2];
          start = (E.start & 4u) != 0;
        }
        Row<VPL> cur = sgm_step<VPL, true>(pf, c, read_lane(pen, s), X);
        if ((s == 0 || PHASE >= 2) && start) cur = c;
        LF[s] = cur;
        pf = cur;
      }
    }
    {  // I chain, slot 7 - it
      const int s = kT - 1 - it;
      if (s < nT) {
        const Row<VPL> c = lds_row<VPL>(Ct, sl.pix[s], d0);
        bool start = false;
        if (s == nT - 1) {
          const bool seg_b = PHASE >= 2 && s >= nA;
          if constexpr (PHASE >= 2) {
            if (seg_b) pi = E.e[3];
          }
          start = (E.start & (seg_b ? 8u : 2u)) != 0;
        } else if (PHASE >= 2 && s == nA - 1) {  // back in the first diagonal
          pi = E.e[1];
          start = (E.start & 2u) != 0;
        }
        Row<VPL> cur = sgm_step<VPL, true>(pi, c, read_lane(pen, 8 + s), X);
        if (start) cur = c;
        LI[s] = cur;
        pi = cur;
      }
    }
  }
}

// The phase's update of the sum tile: S = ((S + [I x pre]) + F) + I x count per slot.
template <int VPL, int PHASE, int FINAL>
__device__ __forceinline__ void tile_add(const PhaseGeom &g, const Slots<PHASE> &sl, const Row<VPL> (&LF)[kT],
                                         const Row<VPL> (&LI)[kT], float *St, const TileArgs &P, int x0,
                                         int y0, float *__restrict__ out, int32_t *__restrict__ disp,
                                         int lane) {
  constexpr int D = 64 * VPL;
  const int d0 = lane * VPL;
  const int nA = sl.nA, nT = sl.nT;
  if (nT == 0) return;
  const Counts ca = seg_counts(PHASE, g.a, x0, y0, P.G, P.true_paths);
  const Counts cb = seg_counts(PHASE, g.b, x0, y0, P.G, P.true_paths);
#pragma unroll
  for (int s = 0; s < kT; ++s) {
    if (s < nT) {
      const int pix = sl.pix[s];
      Row<VPL> acc;
      if (PHASE == 0 && !P.accumulate) {
#pragma unroll
        for (int j = 0; j < VPL; ++j) acc.v[j] = 0.0f;
      } else {  // (STM_SGM_ACCUMULATE: the tile's staging has put the caller's `out` into the sum tile)
        acc = lds_srow<VPL>(St, pix, d0);
      }
      if (PHASE == 2) {
        const int pre = s < nA ? ca.pre : cb.pre;
        for (int r = 0; r < pre; ++r) {
#pragma unroll
          for (int j = 0; j < VPL; ++j) acc.v[j] = acc.v[j] + LI[s].v[j];
        }
      }
#pragma unroll
      for (int j = 0; j < VPL; ++j) acc.v[j] = acc.v[j] + LF[s].v[j];
      if (PHASE < 2) {
#pragma unroll
        for (int j = 0; j < VPL; ++j) acc.v[j] = acc.v[j] + LI[s].v[j];
      } else {
        const int inv = s < nA ? ca.inv : cb.inv;
        for (int r = 0; r < inv; ++r) {
#pragma unroll
          for (int j = 0; j < VPL; ++j) acc.v[j] = acc.v[j] + LI[s].v[j];
        }
      }
      // The last pair's result leaves as the finished sum and / or stays in the sum tile for the
      // winner-takes-all that follows the tile's last barrier (tile_wta_lds).
      if (PHASE < 3 || FINAL != 0) store_row<VPL, true>(St + (size_t)pix * st_stride(D), 0, D, d0, acc);
      if (PHASE == 3 && FINAL != 2) {
        store_row<VPL, true>(out, (size_t)tile_out_pixel(P.G, x0, y0, pix), D, d0, acc);
      }
    }
  }
}

// The same two chains for the common case -- a full 8 x 8 tile, no path beginning inside it
// (E.start == 0) -- without the per-slot validity and image-border logic: a wave issues in
// order and the tile kernel is bound by instruction issue, so what the generic form spends on
// selects, branches and register copies per slot is tile time.  All eight slots exist; on the
// diagonal pairs the second segment begins at slot nA (nA = 8: there is none).
template <int VPL, int PHASE>
__device__ __forceinline__ void tile_chains_fast(const Slots<PHASE> &sl, const Entries<VPL, PHASE> &E,
                                                 bool inv_a, bool inv_b, const float *Ct, const float *Ip, float p1, float p2,
                                                 StepCtx &X, int lane, Row<VPL> (&LF)[kT], Row<VPL> (&LI)[kT]) {
  constexpr int dx = PhaseDir<PHASE>::dx, dy = PhaseDir<PHASE>::dy;
  const int d0 = lane * VPL;
  const int nA = sl.nA;
  float pen;
  {
    const int s = lane & 7;
    const bool inv = (lane >> 3) & 1;
    int pix = sl.pix[0];
#pragma unroll
    for (int k = 1; k < kT; ++k) pix = s == k ? sl.pix[k] : pix;
    const int px = pix & (kT - 1), py = pix >> 3;
    const int qx = px + (inv ? dx : -dx), qy = py + (inv ? dy : -dy);
    pen = sgm_penalty(Ip[(py + 1) * (kT + 2) + px + 1], Ip[(qy + 1) * (kT + 2) + qx + 1], p1, p2);
  }
  // The cost row of the next slot is requested before the current step runs: with sixteen
  // waves on the LDS a read takes a few hundred cycles, which a step issued behind it would
  // wait out once per slot.
  auto step = [&](const Row<VPL> &prev, const Row<VPL> &c, float p2a) {
    if constexpr (VPL >= 2) return sgm_step_fused<VPL>(prev, c, p2a, X);
    else return sgm_step<VPL, true>(prev, c, p2a, X);
  };
  Row<VPL> pf = E.e[0];
  Row<VPL> cn = lds_row<VPL>(Ct, sl.pix[0], d0);
#pragma unroll
  for (int s = 0; s < kT; ++s) {
    if constexpr (PHASE >= 2) {
      if (s > 0 && s == nA) pf = E.e[2];
    }
    const Row<VPL> c = cn;
    cn = lds_row<VPL>(Ct, sl.pix[s < kT - 1 ? s + 1 : kT - 1], d0);  // after slot 7: slot 7 again, for I
    pf = step(pf, c, read_lane(pen, s));
    LF[s] = pf;
  }
  Row<VPL> pi;
  if constexpr (PHASE >= 2) pi = nA < kT ? E.e[3] : E.e[1];
  else pi = E.e[1];
#pragma unroll
  for (int s = kT - 1; s >= 0; --s) {
    if constexpr (PHASE >= 2) {
      if (s < kT - 1 && s == nA - 1) pi = E.e[1];
    }
    const Row<VPL> c = cn;
    if (s > 0) cn = lds_row<VPL>(Ct, sl.pix[s - 1], d0);
    // A segment whose inverse pass the reference never runs (inv_a / inv_b false) is skipped.
    if (PHASE < 2 || (s < nA ? inv_a : inv_b)) {
      pi = step(pi, c, read_lane(pen, 8 + s));
      LI[s] = pi;
    }
  }
}

// ... and the sum update when every direction counts once: S = (S + F) + I.
template <int VPL, int PHASE, int FINAL>
__device__ __forceinline__ void tile_add_fast(const Slots<PHASE> &sl, const Row<VPL> (&LF)[kT],
                                              const Row<VPL> (&LI)[kT], bool inv_a, bool inv_b, float *St, const TiledGeom &G, bool accumulate, int x0, int y0,
                                              float *__restrict__ out, int32_t *__restrict__ disp, int lane) {
  constexpr int D = 64 * VPL;
  const int d0 = lane * VPL;
  const bool from_tile = PHASE != 0 || accumulate;  // (accumulating: pair 0 starts from the staged `out`)
  Row<VPL> sn;
  if (from_tile) sn = lds_srow<VPL>(St, sl.pix[0], d0);
#pragma unroll
  for (int s = 0; s < kT; ++s) {
    const int pix = sl.pix[s];
    Row<VPL> acc;
    if (!from_tile) {
#pragma unroll
      for (int j = 0; j < VPL; ++j) acc.v[j] = 0.0f;
    } else {
      acc = sn;
      if (s < kT - 1) sn = lds_srow<VPL>(St, sl.pix[s + 1], d0);  // the next slot's sum, early
    }
#pragma unroll
    for (int j = 0; j < VPL; ++j) acc.v[j] = acc.v[j] + LF[s].v[j];
    if (PHASE < 2 || (s < sl.nA ? inv_a : inv_b)) {
#pragma unroll
      for (int j = 0; j < VPL; ++j) acc.v[j] = acc.v[j] + LI[s].v[j];
    }
    if (PHASE < 3 || FINAL != 0) store_row<VPL, true>(St + (size_t)pix * st_stride(D), 0, D, d0, acc);
    if (PHASE == 3 && FINAL != 2) {
      store_row<VPL, true>(out, (size_t)tile_out_pixel(G, x0, y0, pix), D, d0, acc);
    }
  }
}

// One pair of a tile for one wave: chains, then (after `between`, the barrier the pair's
// update has to wait for, if any) the update of the sum tile.  Picks the fast forms when the
// tile is full, no path begins in it and the pair counts once everywhere.
template <int VPL, int PHASE, int FINAL, typename Between>
__device__ __forceinline__ void tile_pair(const PhaseGeom &g, const Entries<VPL, PHASE> &E, bool full,
                                          const float *Ct, float *St, const float *Ip, const TileArgs &P,
                                          int x0, int y0, float *__restrict__ out,
                                          int32_t *__restrict__ disp, StepCtx &X, int lane,
                                          Between between, Row<VPL> *exit_f = nullptr) {
  Row<VPL> LF[kT], LI[kT];
  Slots<PHASE> sl;
  sl.init(g);
  // Fast forms: full tile, no path begins in it, every inverse pass counts once or not at all
  // (the reference skips the inverse diagonals of 17-44 % of a wide image's pixels, Q6: those
  // segments' I chains are not even computed); what is left for the generic forms are the
  // tiles on the image border and on the re-walked corner diagonals.
  bool fast = full && E.start == 0;
  bool inv_a = true, inv_b = true;
  if (PHASE >= 2 && fast) {
    const Counts ca = seg_counts(PHASE, g.a, x0, y0, P.G, P.true_paths);
    const Counts cb = seg_counts(PHASE, g.b, x0, y0, P.G, P.true_paths);
    fast = ca.inv <= 1 && ca.pre == 0 && cb.inv <= 1 && cb.pre == 0;
    inv_a = ca.inv != 0;
    inv_b = cb.inv != 0;
  }
  if (fast) {
    tile_chains_fast<VPL, PHASE>(sl, E, inv_a, inv_b, Ct, Ip, P.p1, P.p2, X, lane, LF, LI);
    if (PHASE == 0 && exit_f) *exit_f = LF[kT - 1];
    between();
    tile_add_fast<VPL, PHASE, FINAL>(sl, LF, LI, inv_a, inv_b, St, P.G, P.accumulate != 0, x0, y0, out, disp, lane);
  } else {
    tile_chains<VPL, PHASE>(sl, E, Ct, Ip, P.p1, P.p2, X, lane, LF, LI);
    if (PHASE == 0 && exit_f) *exit_f = LF[kT - 1];  // (the row's rightward exit where it spans the tile: nT == 8)
    between();
    tile_add<VPL, PHASE, FINAL>(g, sl, LF, LI, St, P, x0, y0, out, disp, lane);
  }
}

// Sixteen waves per tile in two groups of eight: group 0 handles the direction pairs 0
// (right/left) and 2 (down-right/up-left), group 1 the pairs 1 (down/up) and 3
// (down-left/up-right).  The chains of a pair only read the cost tile, so the two groups run
// them at the same time; only the four updates of the sum tile are serialised (barriers B1-B3),
// in the reference's order:
//   group 0: chains 0, add 0 | B1 | chains 2        | B2 | add 2    | B3 |
//   group 1: chains 1        | B1 | add 1           | B2 | chains 3 | B3 | add 3
// The tile's cost and sum occupy 128 of the CU's 160 KB of LDS, so one workgroup per CU; it is
// persistent and walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...  A tile's 160 KB of loads
// would be exposed once per tile, so they are issued where the issuing group is about to wait
// anyway and few registers are live (the kernel runs at the 128-register limit of a
// 1024-thread workgroup: both chains' results of a pair are 64 of them):
//   group 0, after add 2 (then waits at B3 for group 1's chains 3): the whole cost tile of the
//            next tile (8 pixels per wave), its intensities and group 0's first border states;
//   group 1, after chains 3: its first border states of the next tile;
//   either group, after its first add: the border states of its second pair.

}  // namespace
}  // namespace stm
