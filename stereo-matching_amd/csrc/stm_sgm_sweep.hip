// Semiglobal aggregation, tiled form: the row sweeps (see stm_sgm_tiled.hip for the whole form).
//
// Reference semantics: src/semiglobal.cpp:116-159 (recurrence), :18-91 (path table): the three
// directions that advance one image row per step (down, down-right, down-left; or the three
// upward ones) computed with ONE read of each cost row; only the states where a path leaves an
// 8 x 8 tile are kept (border states), nothing is added into a sum here.
#include <stdlib.h>

#include <atomic>

#include <type_traits>

#include "stm_sgm_tiled_dev.h"

namespace stm {
namespace {

// ---- granule mailboxes between column strips ---------------------------------------
// One mailbox carries one pixel's D values of one diagonal direction from the edge wave of a
// strip to the edge wave of its neighbour, once per image row.  A value and the index of the
// row it belongs to (+1, never 0) form one naturally aligned 8-byte granule, written by one
// write-through store: the data is its own flag, the consumer re-reads its granules until
// every tag matches (no fence on either side).  kMboxSlots = 4 rows of history are enough:
// a strip that publishes row r has consumed its neighbour's row r-2 and that neighbour is
// therefore at least in row r-2, where it may still need rows r-3.. of this strip.
__device__ __forceinline__ u64 *mbox_ptr(float *ws, const TiledGeom &G, int sweep, int b, int strip,
                                         int side) {
  return reinterpret_cast<u64 *>(ws + G.mbox_off) + mbox_granule_off(G, sweep, b, strip, side);
}

// Two granules travel in one 16-byte write-through access where a lane holds an even number
// of values (each 8-byte half is still written and read whole: MI355X guide, Guideline 16,
// form R2): 8-byte stores reach the memory side as one partial write each, which the sweeps'
// two mailboxes per strip edge and row turned into a tenth of the stage's write traffic.
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
constexpr int kAuxSc1 = 16;  // cache-policy bits of the raw buffer builtins: sc1 (write-through / L2-served)
// The polling loads are re-issued on every trip of a wait loop because the loop carries a
// compiler memory barrier (mbox_wait); tests/test_abi.py checks the generated code for it.
// (Marking the accesses volatile for the compiler -- aux bit 31 -- would do the same, but the
// backend then also widens their scope to sc0 sc1, system scope.)

template <int VPL>
__device__ __forceinline__ void mbox_publish(u64 *box, int D, int d0, unsigned row, const Row<VPL> &r) {
  u64 *slot = box + mbox_slot_granule(D, row) + d0;
  const unsigned tag = row + 1;
  if constexpr (VPL % 2 == 0) {
    const BufRsrc rs = make_rsrc(slot - d0, (unsigned)(D * sizeof(u64)));
#pragma unroll
    for (int j = 0; j < VPL; j += 2) {
      v4u_t q;
      q.x = __builtin_bit_cast(unsigned, r.v[j]); q.y = tag;
      q.z = __builtin_bit_cast(unsigned, r.v[j + 1]); q.w = tag;
      __builtin_amdgcn_raw_buffer_store_b128(q, rs, (d0 + j) * 8, 0, kAuxSc1);
      asm volatile("s_nop 3" ::"v"(q) : "memory");  // (hold_store_data: every 16-byte buffer store, whatever its soffset)
    }
  } else {
#pragma unroll
    for (int j = 0; j < VPL; ++j)
      __hip_atomic_store(slot + j, ((u64)tag << 32) | (u64)__builtin_bit_cast(unsigned, r.v[j]),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// What mbox_peek brings: the raw granules of one row of a mailbox.
template <int VPL>
struct Granules {
  unsigned val[VPL], tag[VPL];
};
// Issues the loads of one polling pass (their results may be looked at much later).
template <int VPL>
__device__ __forceinline__ void mbox_peek(const u64 *box, int D, int d0, unsigned row, Granules<VPL> &q) {
  const u64 *slot = box + mbox_slot_granule(D, row) + d0;
  if constexpr (VPL % 2 == 0) {
    const BufRsrc rs = make_rsrc(slot - d0, (unsigned)(D * sizeof(u64)));
#pragma unroll
    for (int j = 0; j < VPL; j += 2) {
      const v4u_t g = __builtin_amdgcn_raw_buffer_load_b128(rs, (d0 + j) * 8, 0, kAuxSc1);
      q.val[j] = g.x; q.tag[j] = g.y;
      q.val[j + 1] = g.z; q.tag[j + 1] = g.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      const u64 g = __hip_atomic_load(const_cast<u64 *>(slot) + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      q.val[j] = (unsigned)g;
      q.tag[j] = (unsigned)(g >> 32);
    }
  }
}
// True when every lane's granules carry the tag of `row`; r then holds the values.
template <int VPL>
__device__ __forceinline__ bool mbox_take(const Granules<VPL> &q, unsigned row, Row<VPL> &r) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    ok &= q.tag[j] == row + 1;
    r.v[j] = __builtin_bit_cast(float, q.val[j]);
  }
  return __ballot(ok) == ~0ull;
}
// One polling pass.
template <int VPL>
__device__ __forceinline__ bool mbox_try(const u64 *box, int D, int d0, unsigned row, Row<VPL> &r) {
  Granules<VPL> q;
  mbox_peek<VPL>(box, D, d0, row, q);
  return mbox_take<VPL>(q, row, r);
}

// Waits for the neighbour's row `row`.  Bounded: after ~2 s (s_memrealtime runs at 100 MHz)
// the status word is set and this wave never waits again (`dead`), so the grid always drains.
template <int VPL>
__device__ __forceinline__ void mbox_wait(const u64 *box, int D, int d0, unsigned row, Row<VPL> &r,
                                          bool &dead, unsigned *status) {
  if (dead) return;
  if (mbox_try<VPL>(box, D, d0, row, r)) return;
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    __builtin_amdgcn_s_sleep(2);
    asm volatile("" ::: "memory");  // (and nothing the loop reads is carried over from the last trip)
    if (mbox_try<VPL>(box, D, d0, row, r)) return;
    // Somebody's wait has expired already: the call's results are invalid, nobody waits any
    // longer (a failed call drains in ~2 s, not in 2 s per row that is still ahead).
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
      dead = true;
      return;
    }
    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
      dead = true;
      if ((threadIdx.x & 63) == 0) atomicOr(status, STM_SGM_STATUS_TIMEOUT);
      return;
    }
  }
}

// =====================================================================================
// Row sweep: three directions per read of the cost volume.
//
// DOWN: V = down (0,+1), A = down-right (+1,+1), B = down-left (-1,+1).
// UP:   V = up (0,-1),   A = up-right (+1,-1),   B = up-left (-1,-1).
// A moves one column to the right per row (its source is the column on the left), B one to
// the left.  A wave keeps the three states of its CW columns in registers; in row t it needs
// A of the column left of its first one and B of the column right of its last one from row
// t-1: its neighbour waves put them into LDS (double-buffered, one barrier per row), the
// neighbour workgroups into a mailbox.  What leaves the wave is computed and published first
// in every row, what arrives is consumed last, so a hand-off has a whole row of slack.
// =====================================================================================
struct SweepArgs {
  TiledGeom G;
  float p1, p2;
  int dlim;
  int b0;  // first image of this launch (blockIdx.y counts from it)
  long long status_off;
  unsigned long long *stamps;  // diagnostic builds only (stm_sgm_tiled_diag.h), else null
};

// The exchange slots of a sweep workgroup: [parity][A/B][wave or half-wave][d].  At D = 128 the
// interior waves step two columns per half-wave (sgm_step_h2x3): the unit of the exchange is then
// the half-wave, and the other waves follow that numbering.
template <int VPL>
struct SweepXch {
  static constexpr int NV = VPL == 2 ? 2 * kSweepWaves : kSweepWaves;
  float v[2][2][NV][64 * VPL];
};

// One image's sweep for this workgroup's column strip.
template <int VPL, int CW, bool UP>
__device__ __forceinline__ void sweep_image(const float *__restrict__ cost_all, const float *__restrict__ image_all,
                                            float *__restrict__ ws, const SweepArgs &P, const int b,
                                            float (&xch)[2][2][SweepXch<VPL>::NV][64 * VPL], bool &dead) {
  static_assert(CW >= 2, "a wave's leaving states must not depend on its arriving ones");
  constexpr int PF = 2;  // cost rows in flight beyond the current one
  constexpr int NW = kSweepWaves;
  constexpr int D = 64 * VPL;
  constexpr bool H2 = VPL == 2;
  constexpr int NV = SweepXch<VPL>::NV;
  const TiledGeom &G = P.G;
  const int lane = threadIdx.x & 63;
  const int wv = wave_uniform((int)(threadIdx.x >> 6));
  const int s = blockIdx.x;
  const int H = G.H, W = G.W;
  const int d0 = lane * VPL;
  const int xa = sweep_first_col(s, wv, CW);  // this wave's first column
  const bool active = xa < W;
  const size_t npix = (size_t)H * W;
  const float *cost = cost_all + (size_t)b * npix * D;
  const float *image = image_all + (size_t)b * npix;
  unsigned *status = reinterpret_cast<unsigned *>(ws + P.status_off);
  StepCtxN<3> X;
  X.init(P.p1, lane == 63 && P.dlim < D);

  // Neighbours.  Left source of column xa: none at the image border (the path starts there).
  const bool has_left = xa > 0, left_lds = wv > 0;
  const bool has_right = xa + CW < W, right_lds = wv < NW - 1;
  const int sweep = UP ? 1 : 0;
  u64 *pubA = (has_right && !right_lds) ? mbox_ptr(ws, G, sweep, b, s, 0) : nullptr;
  u64 *pubB = (has_left && !left_lds) ? mbox_ptr(ws, G, sweep, b, s, 1) : nullptr;
  const u64 *subA = (has_left && !left_lds) ? mbox_ptr(ws, G, sweep, b, s - 1, 0) : nullptr;
  const u64 *subB = (has_right && !right_lds) ? mbox_ptr(ws, G, sweep, b, s + 1, 1) : nullptr;
  // Exchange slots: where this wave's leaving A / B go and where its arriving ones come from.
  const int slot_pubA = H2 ? 2 * wv + 1 : wv, slot_pubB = H2 ? 2 * wv : wv;
  const int slot_subA = H2 ? 2 * wv - 1 : wv - 1, slot_subB = H2 ? 2 * wv + 2 : wv + 1;

  // A wave at the right border may own fewer than CW columns and re-reads its last one for the
  // others; steps past the last row re-read it (sweep_cost_in_image clamps both).
  auto row_of = [&](int t) { return sweep_row(H, UP, t); };
  unsigned xoff[CW];  // (= sweep_cost_in_image(G, UP, t, xa + j) - the row's term: 32-bit, wave-uniform)
#pragma unroll
  for (int j = 0; j < CW; ++j) xoff[j] = sweep_cost_col_in_row(G, xa + j);
  auto fetch = [&](int t, Row<VPL> (&dst)[CW]) {
    const unsigned base = (unsigned)wave_uniform((int)sweep_cost_row_in_image(G, UP, t));
#pragma unroll
    for (int j = 0; j < CW; ++j) dst[j] = load_row_at<VPL, true, true>(cost, (long)(base + xoff[j]), D, d0);
  };
  // Intensities for the 3*CW penalties of a row: lane l serves column l/3, direction l%3
  // (0 V, 1 A, 2 B) and reads its pixel and that pixel's predecessor on the path.
  const int pj = imin(lane / 3, CW - 1), pk = lane % 3;
  const int px = imin(xa + pj, W - 1);
  const int pxs = sweep_pred_col(px, pk, W);
  auto fetch_intensity = [&](int t, float &icur, float &iprev) {
    icur = image[sweep_intensity_cur(G, UP, t, px)];
    iprev = image[sweep_intensity_prev(G, UP, t, pxs)];
  };

  // Where this wave's column states go: at most one column with x % 8 == 7 (A) and one with
  // x % 8 == 0 (B) per eight columns.
  Row<VPL> V[CW], A[CW], B[CW];
  // Vector-memory results come back in issue order, so every row issues its loads in the
  // order in which later rows consume them: [intensities and cost rows of row t + PF + 1]
  // [at the row's end: what the neighbour strips have published for the next row].
  Row<VPL> cb[PF + 1][CW];
  float ic[PF + 1], ip[PF + 1];
  Granules<VPL> qa, qb;
  Granules<4> qa4;  // (the half-wave form's: four disparities per lane, one mailbox per wave)

  // The common case gets a loop of its own: every column of the wave inside the image and none
  // of them in image column 0 or W-1, so no path starts here and nothing is clipped.  At
  // D <= 128 the sweep is bound by instruction issue (16 waves x ~800 instructions per row of the
  // generic loop below), and what that loop spends on start / validity selects and on
  // recomputing 64-bit addresses every row is row time; at D = 256 the sweep is HBM-bound and
  // the two loops take the same time.
  if constexpr (H2) {
    constexpr int PFH0 = 2;
    if (active) {
      static_assert(!H2 || CW == 4, "two columns per half-wave");
      // ---- D = 128: lanes 0-31 own columns xa, xa + 1, lanes 32-63 columns xa + 2, xa + 3, four
      // disparities per lane; a row is two batches of three steps.  A wave at an image border
      // (`edge`: it holds column 0, column W - 1 or columns past it) additionally starts the
      // paths that begin there (L = C), re-reads its last valid column for the missing ones and
      // keeps their states to itself; a workgroup -- and through the hand-offs the whole sweep --
      // runs at the pace of its slowest wave, which the one-pixel-per-wave loop below made the
      // edge waves.
      const bool edge = !(xa >= 1 && xa + CW <= W - 1);
      const int half = lane >> 5, hl = lane & 31, dh = hl * 4;
      const int vw = 2 * wv + half;
      constexpr int PFH = PFH0;  // cost rows in flight beyond the current one
      // Addresses are a wave-uniform base plus this lane's 32-bit offset (its half's first
      // column and its disparities).
      const unsigned loff = (unsigned)(half * 2 * D + dh);
      const int col0 = xa + 2 * half;  // this half's columns: col0, col0 + 1
      const unsigned lload0 = (unsigned)((imin(col0, W - 1) - xa) * D + dh);
      const unsigned lload1 = (unsigned)((imin(col0 + 1, W - 1) - xa) * D + dh);
      const bool valid0 = col0 < W, valid1 = col0 + 1 < W;
      const bool start_a0 = col0 == 0;                             // a rightward path begins in image column 0
      const bool start_b0 = col0 >= W - 1, start_b1 = col0 + 1 >= W - 1;  // a leftward one in column W - 1
      SweepCursor rows;  // the next row to fetch
      rows.init(G, UP, xa);
      // Penalties: lane i of every 16-lane row serves column i / 3 (0, 1) of the row's half and
      // direction i % 3 (0 V, 1 A, 2 B).
      const int i16 = lane & 15;
      const int pxh = imin(col0 + imin(i16 / 3, 1), W - 1), pkh = i16 % 3;
      const int pxsh = sweep_pred_col(pxh, pkh, W);
      Row<4> Vh[2], Ah[2], Bh[2], ch[PFH + 1][2];
      float ich[PFH + 1], iph[PFH + 1];
      auto refill = [&](Row<4> (&dst)[2], float &icur, float &iprev) {
        icur = (image + rows.cur_row())[(unsigned)pxh];
        iprev = (image + rows.prev_row(G))[(unsigned)pxsh];
        const float *cnext = cost + rows.cost_row();
        dst[0] = load_row<4, true, true>(cnext + lload0, 0, D, 0);
        dst[1] = load_row<4, true, true>(cnext + lload1, 0, D, 0);
        rows.advance(G);  // (past the end: the last row again, unused)
      };
      const bool colA_here = (xa & (kT - 1)) == 4, colB_here = (xa & (kT - 1)) == 0;  // x % 8 == 7 is xa + 3, == 0 is xa
      const int y_first = UP ? H - 1 : 0;
      float *colA = colA_here && xa + 3 < W ? colb_ptr(ws, G, b, UP ? 2 : 1, (xa + 3) >> 3, y_first) : nullptr;
      float *colB = colB_here ? colb_ptr(ws, G, b, UP ? 5 : 4, xa >> 3, y_first) : nullptr;
      float *rowV = rowb_ptr(ws, G, b, UP ? 3 : 0, 0, xa);  // line 0; line t is t * rowb_line_stride further
      const long long kstride = rowb_dir_stride(G);
      const float p1v = in_vgpr(P.p1);
      float below[3], above[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) below[k] = above[k] = in_vgpr(__builtin_inff());
      const bool drop_tail = hl == 31 && P.dlim < D;
      // LDS exchange, in floats from xch[0][0][0][0]: own slots and the neighbours'.
      float *xbase = &xch[0][0][0][0];
      constexpr int kAB = NV * D, kPar = 2 * NV * D;  // from A to B, from parity 0 to 1
      const unsigned my_off = (unsigned)(vw * D + dh);
      const unsigned inA_off = (unsigned)((vw > 0 ? vw - 1 : 0) * D + dh);
      const unsigned inB_off = (unsigned)(kAB + (vw < NV - 1 ? vw + 1 : NV - 1) * D + dh);
      // Mailbox role of the wave: 0 none (interior waves, and edge waves at an image border),
      // 1 first wave of a strip with a neighbour on the left (publishes B of its first column
      // from half 0, takes A for it), 2 last wave with a neighbour on the right (publishes A of
      // its last column from half 1, takes B).  Both only in a one-wave strip, which the sweeps
      // never have.  The row loop is instantiated per role so that what a wave issues to memory
      // per row is the same on (nearly) every path through the row: a load or store under a
      // wave-uniform condition is a branch around it, behind which hipcc no longer knows how
      // many accesses are in flight and waits for ALL of them at the row's first use of a cost
      // row (s_waitcnt vmcnt(0): the two rows of prefetch included, once per row; 51 % of the
      // sweep's time was VALU work, 63 % waiting).  Column-line stores go through buffer
      // descriptors for the same reason: the half with nothing to store stores out of range.
      const int role = (pubB && subA) ? 1 : ((pubA && subB) ? 2 : 0);
      constexpr unsigned kNowhere = 0x7fffffffu;
      // A column line's rows from row 0 on (colA / colB point at row y_first; no line: an empty
      // range); the row in flight is the scalar part of the address.  (Two stores rather than one
      // with selected operands: a select between Ah[1] and Bh[0] sends both arrays to scratch.)
      const BufRsrc colA_rs = make_rsrc(colA ? colA - (long long)y_first * colb_row_stride(G) : ws, colA ? colb_line_bytes(G) : 0u);
      const BufRsrc colB_rs = make_rsrc(colB ? colB - (long long)y_first * colb_row_stride(G) : ws, colB ? colb_line_bytes(G) : 0u);
      // (Per-lane offsets of the conditional accesses are formed where they are used, from a
      // laundered copy of the lane's half: hoisted out of the row loop each of them is a register
      // the loop does not have -- it runs at the 128 a 16-wave workgroup can have.)
      auto my_half = [&]() {
        unsigned h = (unsigned)half;
        asm volatile("" : "+v"(h));
        return h;
      };
      auto run = [&](auto role_tag) {
        constexpr int ROLE = decltype(role_tag)::value;
        const u64 *sub = ROLE == 1 ? subA : subB;  // (ROLE != 0)
        u64 *pub = ROLE == 1 ? pubB : pubA;
        const BufRsrc pub_rs = make_rsrc(ROLE != 0 ? (const void *)pub : (const void *)ws, ROLE != 0 ? mbox_bytes(D) : 0u);
        const BufRsrc sub_rs = make_rsrc(ROLE != 0 ? (const void *)sub : (const void *)ws, ROLE != 0 ? mbox_bytes(D) : 0u);
        auto leave = [&](int par, unsigned t) {  // what leaves the half: A of its last column, B of its first
          if constexpr (ROLE != 0) {
            const Row<4> &x = ROLE == 2 ? Ah[1] : Bh[0];
            const unsigned tag = t + 1;
            const unsigned soff = mbox_slot_byte(D, t);
            if (half == (ROLE == 2 ? 1 : 0)) {
#pragma unroll
              for (int j = 0; j < 4; j += 2) {
                v4u_t q;
                q.x = __builtin_bit_cast(unsigned, x.v[j]); q.y = tag;
                q.z = __builtin_bit_cast(unsigned, x.v[j + 1]); q.w = tag;
                __builtin_amdgcn_raw_buffer_store_b128(q, pub_rs, (unsigned)dh * 8u, soff + (unsigned)j * 8u, kAuxSc1);
                asm volatile("s_nop 3" ::"v"(q) : "memory");  // (hold_store_data: the store is still reading q)
              }
            }
          }
          store_row<4, true>(xbase + par * kPar + my_off, 0, D, 0, Ah[1]);
          store_row<4, true>(xbase + par * kPar + kAB + my_off, 0, D, 0, Bh[0]);
        };
        auto peek = [&](unsigned t) {  // requests the neighbour's row t (looked at a row later)
          if constexpr (ROLE != 0) {
            const unsigned soff = mbox_slot_byte(D, t);
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
              const v4u_t g = __builtin_amdgcn_raw_buffer_load_b128(sub_rs, (unsigned)dh * 8u, soff + (unsigned)j * 8u, kAuxSc1);
              qa4.val[j] = g.x; qa4.tag[j] = g.y;
              qa4.val[j + 1] = g.z; qa4.tag[j + 1] = g.w;
            }
          }
        };

#pragma unroll
        for (int u = 0; u <= PFH; ++u) refill(ch[u], ich[u], iph[u]);
        // Row 0, then whole groups of PFH + 1 rows, every wave of the sweep alike
        // (sweep_rows_run): the rows past the last one step on the last row's costs again,
        // exchange and publish like any other (their neighbours run them too) and store nothing
        // -- a test `t < H` (or `t == 0`) around a row is one more join at which the count of
        // accesses in flight is lost.
        int y = y_first;
        auto row = [&](const int t, auto u_tag, auto first_tag) {
            constexpr int u = decltype(u_tag)::value;
            constexpr bool FIRST = decltype(first_tag)::value;
            const bool real = t < H;
            {
              const int par = t & 1;
              const float pen = sgm_penalty(ich[u], iph[u], P.p1, P.p2);
              if constexpr (FIRST) {
#pragma unroll
                for (int j = 0; j < 2; ++j) { Vh[j] = ch[u][j]; Ah[j] = ch[u][j]; Bh[j] = ch[u][j]; }
                leave(par, 0u);
              } else {
                {  // V in column 0, A into column 1 (from column 0), B into column 0 (from column 1)
                  Row<4> st[3] = {Vh[0], Ah[0], Bh[1]};
                  const Row<4> cc[3] = {ch[u][0], ch[u][1], ch[u][0]};
                  sgm_step_h2x3<0, 4, 2>(st, cc, pen, p1v, below, above, drop_tail);
                  Vh[0] = st[0];
                  Ah[1] = st[1];
                  Bh[0] = st[2];
                  if (edge) Bh[0] = pick_row(start_b0, ch[u][0], Bh[0]);
                }
                leave(par, (unsigned)t);
                {  // V in column 1, A into column 0 and B into column 1 from what arrived
                  Row<4> st[3];
                  st[0] = Vh[1];
                  st[1] = load_row<4, true>(xbase + (par ^ 1) * kPar + inA_off, 0, D, 0);
                  st[2] = load_row<4, true>(xbase + (par ^ 1) * kPar + inB_off, 0, D, 0);
                  if constexpr (ROLE != 0) {
                    Row<4> r;
                    if (dead || !mbox_take<4>(qa4, (unsigned)(t - 1), r))
                      mbox_wait<4>(sub, D, dh, (unsigned)(t - 1), r, dead, status);
                    if constexpr (ROLE == 1) {  // its first half takes A from the strip on the left
                      if (half == 0) st[1] = r;
                    } else {                    // its second half takes B from the strip on the right
                      if (half == 1) st[2] = r;
                    }
                  }
                  const Row<4> cc[3] = {ch[u][1], ch[u][0], ch[u][1]};
                  sgm_step_h2x3<3, 1, 5>(st, cc, pen, p1v, below, above, drop_tail);
                  Vh[1] = st[0];
                  Ah[0] = st[1];
                  Bh[1] = st[2];
                  if (edge) {
                    Ah[0] = pick_row(start_a0, ch[u][0], Ah[0]);
                    Bh[1] = pick_row(start_b1, ch[u][1], Bh[1]);
                  }
                }
              }
              peek((unsigned)t);
              if (real && sweep_stores_row(UP, y)) {
                float *r0 = rowV + (long long)(y >> 3) * rowb_line_stride(G);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                  if (!edge || (j == 0 ? valid0 : valid1)) {
                    store_row<4, true, true>(r0 + (loff + (unsigned)(j * D)), 0, D, 0, Vh[j]);
                    store_row<4, true, true>(r0 + kstride + (loff + (unsigned)(j * D)), 0, D, 0, Ah[j]);
                    store_row<4, true, true>(r0 + 2 * kstride + (loff + (unsigned)(j * D)), 0, D, 0, Bh[j]);
                  }
                }
              }
              {  // the column lines of this wave (A where x % 8 == 7, B where x % 8 == 0), row y
                const int soff = real ? (int)colb_line_row_byte(G, y) : 0;
                const unsigned h = my_half();
                buf_store_row<4>(colA_rs, (int)(real && h == 1 ? (unsigned)dh * 4u : kNowhere), soff, Ah[1]);
                buf_store_row<4>(colB_rs, (int)(real && h == 0 ? (unsigned)dh * 4u : kNowhere), soff, Bh[0]);
              }
              y += UP ? -1 : 1;
              refill(ch[u], ich[u], iph[u]);
              lds_barrier();
            }
        };
        static_assert(PFH == 2, "the rows of a group name their register set");
        row(0, std::integral_constant<int, 0>{}, std::true_type{});
        for (int t0 = 1; t0 < sweep_rows_run(H, PFH); t0 += PFH + 1) {
          row(t0, std::integral_constant<int, 1>{}, std::false_type{});
          row(t0 + 1, std::integral_constant<int, 2>{}, std::false_type{});
          row(t0 + 2, std::integral_constant<int, 0>{}, std::false_type{});
        }
      };
      if (role == 0) run(std::integral_constant<int, 0>{});
      else if (role == 1) run(std::integral_constant<int, 1>{});
      else run(std::integral_constant<int, 2>{});
      return;
    }
    // (a wave past the image only keeps the row barriers)
    for (int t = 0; t < sweep_rows_run(H, PFH0); ++t) lds_barrier();
    return;
  } else if (VPL <= 2 && xa >= 1 && xa + CW <= W - 1) {
    SweepCursor rows;  // the next row to fetch
    rows.init(G, UP, xa);
    // Intensity samples of this lane's penalty (cur: pixel xa + pj of the row, prv: its
    // predecessor on the path, one row back).
    auto refill = [&](Row<VPL> (&dst)[CW], float &icur, float &iprev) {
      icur = (image + rows.cur_row())[px];
      iprev = (image + rows.prev_row(G))[pxs];
      const float *cnext = cost + rows.cost_row() + d0;  // this lane's part of the row
#pragma unroll
      for (int j = 0; j < CW; ++j) dst[j] = load_row<VPL, true, true>(cnext + j * D, 0, D, 0);
      rows.advance(G);  // (past the end: the last row again, unused)
    };
    // Border-state destinations.  Column lines: A where x % 8 == 7, B where x % 8 == 0 (at most
    // one of each among CW <= 8 columns), advancing by one line row per image row.
    int ja_col = -1, jb_col = -1;
#pragma unroll
    for (int j = 0; j < CW; ++j) {
      if (((xa + j) & (kT - 1)) == kT - 1) ja_col = j;
      if (((xa + j) & (kT - 1)) == 0) jb_col = j;
    }
    const int y_first = UP ? H - 1 : 0;
    float *colA = ja_col >= 0 ? colb_ptr(ws, G, b, UP ? 2 : 1, (xa + ja_col) >> 3, y_first) + d0 : nullptr;
    float *colB = jb_col >= 0 ? colb_ptr(ws, G, b, UP ? 5 : 4, (xa + jb_col) >> 3, y_first) + d0 : nullptr;
    const int col_step = UP ? -colb_row_stride(G) : colb_row_stride(G);
    float *rowV = rowb_ptr(ws, G, b, UP ? 3 : 0, 0, xa) + d0;  // line 0; line t is t * rowb_line_stride further
    const long long kstride = rowb_dir_stride(G);

#pragma unroll
    for (int u = 0; u <= PF; ++u) refill(cb[u], ic[u], ip[u]);
    int y = y_first;
    for (int t0 = 0; t0 < H; t0 += PF + 1) {
#pragma unroll
      for (int u = 0; u <= PF; ++u) {
        const int t = t0 + u;
        if (t < H) {
          const int par = t & 1;
          STM_SWEEP_STAMP(0);
          const float pen = sgm_penalty(ic[u], ip[u], P.p1, P.p2);
          if (t == 0) {
#pragma unroll
            for (int j = 0; j < CW; ++j) { V[j] = cb[u][j]; A[j] = cb[u][j]; B[j] = cb[u][j]; }
            if (pubA) mbox_publish<VPL>(pubA, D, d0, 0u, A[CW - 1]);
            if (pubB) mbox_publish<VPL>(pubB, D, d0, 0u, B[0]);
            store_row<VPL, true>(&xch[par][0][slot_pubA][0], 0, D, d0, A[CW - 1]);
            store_row<VPL, true>(&xch[par][1][slot_pubB][0], 0, D, d0, B[0]);
          } else {
#pragma unroll
            for (int i = 0; i < CW; ++i) {
              const int ja = CW - 1 - i, jb = i;
              Row<VPL> prev[3], cc[3], cur[3];
              prev[0] = V[i];
              cc[0] = cb[u][i];
              cc[1] = cb[u][ja];
              cc[2] = cb[u][jb];
              if (i < CW - 1) {
                prev[1] = A[ja - 1];
                prev[2] = B[jb + 1];
              } else {
                STM_SWEEP_STAMP(2);
                if (left_lds) prev[1] = load_row<VPL, true>(&xch[par ^ 1][0][slot_subA][0], 0, D, d0);
                else if (dead || !mbox_take<VPL>(qa, (unsigned)(t - 1), prev[1]))
                  mbox_wait<VPL>(subA, D, d0, (unsigned)(t - 1), prev[1], dead, status);
                if (right_lds) prev[2] = load_row<VPL, true>(&xch[par ^ 1][1][slot_subB][0], 0, D, d0);
                else if (dead || !mbox_take<VPL>(qb, (unsigned)(t - 1), prev[2]))
                  mbox_wait<VPL>(subB, D, d0, (unsigned)(t - 1), prev[2], dead, status);
                STM_SWEEP_STAMP(3);
              }
              const float p2a[3] = {read_lane(pen, 3 * i), read_lane(pen, 3 * ja + 1),
                                    read_lane(pen, 3 * jb + 2)};
              sgm_step_n<VPL, 3>(prev, cc, p2a, X, cur);
              V[i] = cur[0];
              A[ja] = cur[1];
              B[jb] = cur[2];
              if (i == 0) {
                if (pubA) mbox_publish<VPL>(pubA, D, d0, (unsigned)t, A[CW - 1]);
                if (pubB) mbox_publish<VPL>(pubB, D, d0, (unsigned)t, B[0]);
                store_row<VPL, true>(&xch[par][0][slot_pubA][0], 0, D, d0, A[CW - 1]);
                store_row<VPL, true>(&xch[par][1][slot_pubB][0], 0, D, d0, B[0]);
                STM_SWEEP_STAMP(1);
              }
            }
          }
          STM_SWEEP_STAMP(4);
          if (!dead && t + 1 < H) {
            if (subA) mbox_peek<VPL>(subA, D, d0, (unsigned)t, qa);
            if (subB) mbox_peek<VPL>(subB, D, d0, (unsigned)t, qb);
          }
          if (sweep_stores_row(UP, y)) {
            float *r0 = rowV + (long long)(y >> 3) * rowb_line_stride(G);
#pragma unroll
            for (int j = 0; j < CW; ++j) {
              store_row<VPL, true, true>(r0 + j * D, 0, D, 0, V[j]);
              store_row<VPL, true, true>(r0 + kstride + j * D, 0, D, 0, A[j]);
              store_row<VPL, true, true>(r0 + 2 * kstride + j * D, 0, D, 0, B[j]);
            }
          }
#pragma unroll
          for (int j = 0; j < CW; ++j) {
            if (j == ja_col) store_row<VPL, true, true>(colA, 0, D, 0, A[j]);
            if (j == jb_col) store_row<VPL, true, true>(colB, 0, D, 0, B[j]);
          }
          if (colA) colA += col_step;
          if (colB) colB += col_step;
          y += UP ? -1 : 1;
          // This row's cost registers are free now: fetch row t + PF + 1 into them.
          refill(cb[u], ic[u], ip[u]);
          STM_SWEEP_STAMP(5);
          lds_barrier();
          STM_SWEEP_STAMP(6);
        }
      }
    }
    return;
  }

  if (active) {
#pragma unroll
    for (int u = 0; u <= PF; ++u) {
      fetch_intensity(u, ic[u], ip[u]);
      fetch(u, cb[u]);
    }
  }
  for (int t0 = 0; t0 < H; t0 += PF + 1) {
#pragma unroll
    for (int u = 0; u <= PF; ++u) {
      const int t = t0 + u;
      if (t < H) {
        const int y = row_of(t);
        const int par = t & 1;
        STM_SWEEP_STAMP(0);
        if (active) {
          Row<VPL> c[CW];
#pragma unroll
          for (int j = 0; j < CW; ++j) c[j] = cb[u][j];
          const float pen = sgm_penalty(ic[u], ip[u], P.p1, P.p2);
          fetch_intensity(t + PF + 1, ic[u], ip[u]);
          fetch(t + PF + 1, cb[u]);
          if (t == 0) {
#pragma unroll
            for (int j = 0; j < CW; ++j) { V[j] = c[j]; A[j] = c[j]; B[j] = c[j]; }
            if (pubA) mbox_publish<VPL>(pubA, D, d0, 0u, A[CW - 1]);
            if (pubB) mbox_publish<VPL>(pubB, D, d0, 0u, B[0]);
            store_row<VPL, true>(&xch[par][0][slot_pubA][0], 0, D, d0, A[CW - 1]);
            store_row<VPL, true>(&xch[par][1][slot_pubB][0], 0, D, d0, B[0]);
          } else {
            // CW batches of three independent steps: batch i steps A into column CW-1-i (from
            // the column on its left), B into column i (from the column on its right) and V in
            // column i.  In place: every state is read by an earlier batch than the one that
            // overwrites it.  Batch 0 produces what leaves the wave (A of the last column, B of
            // the first) and needs nothing from outside; the last batch consumes what arrives.
#pragma unroll
            for (int i = 0; i < CW; ++i) {
              const int ja = CW - 1 - i, jb = i;
              Row<VPL> prev[3], cc[3], cur[3];
              prev[0] = V[i];
              cc[0] = c[i];
              cc[1] = c[ja];
              cc[2] = c[jb];
              if (i < CW - 1) {
                prev[1] = A[ja - 1];
                prev[2] = B[jb + 1];
              } else {
                STM_SWEEP_STAMP(2);
                // What arrives: A for the first column, B for the last.
                if (has_left) {
                  if (left_lds) prev[1] = load_row<VPL, true>(&xch[par ^ 1][0][slot_subA][0], 0, D, d0);
                  else if (dead || !mbox_take<VPL>(qa, (unsigned)(t - 1), prev[1]))
                    mbox_wait<VPL>(subA, D, d0, (unsigned)(t - 1), prev[1], dead, status);
                } else {
                  prev[1] = c[0];
                }
                if (has_right) {
                  if (right_lds) prev[2] = load_row<VPL, true>(&xch[par ^ 1][1][slot_subB][0], 0, D, d0);
                  else if (dead || !mbox_take<VPL>(qb, (unsigned)(t - 1), prev[2]))
                    mbox_wait<VPL>(subB, D, d0, (unsigned)(t - 1), prev[2], dead, status);
                } else {
                  prev[2] = c[CW - 1];
                }
              }
              if (i == CW - 1) STM_SWEEP_STAMP(3);
              const float p2a[3] = {read_lane(pen, 3 * i), read_lane(pen, 3 * ja + 1),
                                    read_lane(pen, 3 * jb + 2)};
              sgm_step_n<VPL, 3>(prev, cc, p2a, X, cur);
              V[i] = cur[0];
              // A path starts in image column 0, a B path in column W-1 (L = C there).
              A[ja] = (xa + ja == 0) ? c[ja] : cur[1];
              B[jb] = (xa + jb >= W - 1) ? c[jb] : cur[2];
              if (i == 0) {
                if (pubA) mbox_publish<VPL>(pubA, D, d0, (unsigned)t, A[CW - 1]);
                if (pubB) mbox_publish<VPL>(pubB, D, d0, (unsigned)t, B[0]);
                store_row<VPL, true>(&xch[par][0][slot_pubA][0], 0, D, d0, A[CW - 1]);
                store_row<VPL, true>(&xch[par][1][slot_pubB][0], 0, D, d0, B[0]);
                STM_SWEEP_STAMP(1);
              }
            }
          }
          STM_SWEEP_STAMP(4);
          // What the neighbour strips published in THIS row (at its start, a row's work ago)
          // is requested now and looked at at the end of the next row: a poll from a CU that
          // streams cost rows queues behind them for microseconds, a whole row hides that.
          if (!dead && t + 1 < H) {
            if (subA) mbox_peek<VPL>(subA, D, d0, (unsigned)t, qa);
            if (subB) mbox_peek<VPL>(subB, D, d0, (unsigned)t, qb);
          }
          // Border states.  Row lines: every column of this wave; column lines: A where a
          // rightward path leaves a tile (x % 8 == 7), B where a leftward one does (x % 8 == 0).
          if (sweep_stores_row(UP, y)) {
#pragma unroll
            for (int j = 0; j < CW; ++j) {
              if (xa + j < W) {
                float *r0 = rowb_ptr(ws, G, b, UP ? 3 : 0, y >> 3, xa + j);
                const long long kstride = rowb_dir_stride(G);
                store_row<VPL, true, true>(r0, 0, D, d0, V[j]);
                store_row<VPL, true, true>(r0 + kstride, 0, D, d0, A[j]);
                store_row<VPL, true, true>(r0 + 2 * kstride, 0, D, d0, B[j]);
              }
            }
          }
#pragma unroll
          for (int j = 0; j < CW; ++j) {
            const int x = xa + j;
            if (x < W && (x & (kT - 1)) == kT - 1)
              store_row<VPL, true, true>(colb_ptr(ws, G, b, UP ? 2 : 1, x >> 3, y), 0, D, d0, A[j]);
            if (x < W && (x & (kT - 1)) == 0)
              store_row<VPL, true, true>(colb_ptr(ws, G, b, UP ? 5 : 4, x >> 3, y), 0, D, d0, B[j]);
          }
        }
        // One barrier per row: xch[par] written in this row is read in the next one, and
        // xch[par ^ 1] read in this row is rewritten in the next one.
        STM_SWEEP_STAMP(5);
        lds_barrier();
        STM_SWEEP_STAMP(6);
      }
    }
  }
}

// One launch sweeps the images P.b0 .. P.b0 + gridDim.y - 1 (never more workgroups than the
// device has CUs: make_plan).  A persistent form -- one launch per direction, every workgroup
// taking image after image -- was measured and lost 7-12 %: the loop over images costs the
// kernel the few registers it has left, and a spilled value reloaded in the row loop drains
// every cost-row prefetch in flight (s_waitcnt vmcnt(0)) once per row.
template <int VPL, int CW, bool UP>
__global__ __launch_bounds__(kSweepWaves * 64) void sgm_sweep_kernel(const float *__restrict__ cost_all,
                                                                    const float *__restrict__ image_all,
                                                                    float *__restrict__ ws,
                                                                    const SweepArgs P) {
  __shared__ __attribute__((aligned(16))) SweepXch<VPL> xch;
  bool dead = false;  // a bounded wait of this wave has expired: it never waits again
  sweep_image<VPL, CW, UP>(cost_all, image_all, ws, P, P.b0 + (int)blockIdx.y, xch.v, dead);
}

// Launches one row sweep.  Its workgroups wait for each other, so all of them must be resident
// together.  What guarantees that: the grid never has more workgroups than the device has CUs
// (make_plan), one workgroup of this kernel fits a CU (sweep_fits: registers and LDS, asked from
// the runtime's occupancy calculator once per device and D; a call for which it does not hold
// never selects the tiled form), and the CALLER'S CONTRACT that nothing else holds the device's
// CUs for seconds while the call runs (include/stereomatch_hip.h).  Behind that, every wait is
// bounded and a wait that expires is reported in the status word.
//   STM_SGM_COOPERATIVE makes the launch a cooperative one (hipLaunchCooperativeKernel): the
// runtime then validates the grid against the kernel's occupancy itself and runs cooperative
// launches one at a time per device, which makes tiled calls from several streams of one
// process safe beside each other.  The price is why it is not the default: ROCm's runtime takes
// exclusive access to the device for such a launch -- the host thread waits for the stream's
// earlier work before it submits -- ~60 us per sweep launch, 1 ms of a 42 ms Teddy batch, and the
// call is no longer asynchronous.  Not used while the stream is being captured into a graph.
template <typename Kernel>
int launch_sweep(Kernel kernel, dim3 grid, dim3 block, hipStream_t s, bool cooperative, const float *cost,
                 const float *left, float *ws, const SweepArgs &A) {
  if (cooperative) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) (void)hipGetLastError();
    if (cap != hipStreamCaptureStatusNone) cooperative = false;
  }
  if (!cooperative) {
    hipLaunchKernelGGL(kernel, grid, block, 0, s, cost, left, ws, A);
    return check_launch("stm_sgm_f32 (tiled: row sweep)");
  }
  void *args[] = {(void *)&cost, (void *)&left, (void *)&ws, (void *)&A};
  const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(kernel), grid, block, args, 0, s);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(STM_EHIP,
                "stm_sgm_f32 (tiled: row sweep): the device cannot hold the sweep's %u x %u workgroups at once "
                "(%s); use STM_SGM_PAIRED for this call", grid.x, grid.y, hipGetErrorString(e));
  }
  return STM_OK;
}

// Whether one sweep workgroup per CU fits (registers, LDS) -- per device and D, asked once.
template <int VPL>
bool sweep_fits_vpl() {
  constexpr int CW = 8 / VPL;
  int n0 = 0, n1 = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n0, sgm_sweep_kernel<VPL, CW, false>, kSweepWaves * 64, 0) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&n1, sgm_sweep_kernel<VPL, CW, true>, kSweepWaves * 64, 0) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return n0 >= 1 && n1 >= 1;
}
}  // namespace

bool tiled_sweep_fits(int D) {
  // [device][VPL index]: 0 unknown, 1 fits, -1 does not.  Atomic: calls come from any thread; two
  // threads that both find 0 ask twice and store the same answer.
  static std::atomic<signed char> cache[16][3];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  const int k = D == 64 ? 0 : (D == 128 ? 1 : 2);
  if (dev >= 0 && dev < 16) {
    const signed char c = cache[dev][k].load(std::memory_order_relaxed);
    if (c != 0) return c > 0;
  }
  // (hipOccupancyMaxActiveBlocksPerMultiprocessor is a host-side query of the code object: no
  // stream operation, legal while a stream is capturing)
  const bool ok = D == 64 ? sweep_fits_vpl<1>() : (D == 128 ? sweep_fits_vpl<2>() : sweep_fits_vpl<4>());
  if (dev >= 0 && dev < 16) cache[dev][k].store(ok ? 1 : -1, std::memory_order_relaxed);
  return ok;
}


int tiled_launch_sweep(const float *cost, const float *left, float *ws, const TiledCall &c, bool up, hipStream_t s) {
  const TiledGeom &G = c.G;
  for (int b0 = 0; b0 < G.batch; b0 += c.group) {
    SweepArgs A;
    A.G = G; A.p1 = c.p1; A.p2 = c.p2; A.dlim = c.dlim; A.b0 = b0; A.status_off = c.status_off;
    A.stamps = c.sweep_stamps;
    const int n = G.batch - b0 < c.group ? G.batch - b0 : c.group;
    const dim3 grid(G.strips, n), block(kSweepWaves * 64);
    int rc;
#define STM_SWEEP(VPL_)                                                                                              \
  rc = up ? launch_sweep(sgm_sweep_kernel<VPL_, 8 / VPL_, true>, grid, block, s, c.cooperative, cost, left, ws, A)    \
          : launch_sweep(sgm_sweep_kernel<VPL_, 8 / VPL_, false>, grid, block, s, c.cooperative, cost, left, ws, A)
    switch (G.D / 64) {
      case 1: STM_SWEEP(1); break;
      case 2: STM_SWEEP(2); break;
      default: STM_SWEEP(4); break;
    }
#undef STM_SWEEP
    if (rc != STM_OK) return rc;
  }
  return STM_OK;
}

}  // namespace stm
