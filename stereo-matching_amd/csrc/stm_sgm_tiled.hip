// Semiglobal aggregation, tiled form: the cost volume is read five times and the running
// sum never travels.
//
// Reference semantics: src/semiglobal.cpp:18-91 (path table), :116-159 (recurrence),
// :191-194 (for each path: run(path); run(path.inverse())) -- see stm_sgm.hip for the
// restatement.  out(p) = ((((((L> + L<) + Ldown) + Lup) + Ldr) + Lul) + Ldl) + Lur in float,
// in that order, with the reference's quirks for the two inverse diagonal families (Q6).
//
// The paired form (stm_sgm.hip) walks every path per wave: 4 read-only checkpoint walks + 4
// pair launches, each of which reads the cost volume and read-modify-writes the sum: 56-60
// bytes per (pixel, disparity).  Here the volume is cut into 8 x 8-pixel tiles:
//
//   1. BORDER STATES.  For every direction r, L_r is stored at the pixels where a path leaves a
//      tile: rows y = 7 (mod 8) for the three downward directions, y = 0 (mod 8) for the three
//      upward ones, columns x = 7 (mod 8) for the three rightward and x = 0 (mod 8) for the
//      three leftward ones.  12/8 floats per element in all.  They come from
//        * sgm_sweep_kernel, down and up: a row sweep in which a wave owns CW image columns
//          and steps the three directions that advance one row per step (down, down-right,
//          down-left; or the three upward ones) with ONE read of each cost row.  The
//          diagonal states move one column per row: between the waves of a workgroup through
//          LDS (one barrier per row), between workgroups (column strips) through 8-byte
//          {row tag, value} granules in global memory, written write-through and polled by
//          the consumer (MI355X guide, Guideline 16, form R2).  All workgroups of a launch
//          must be resident: the grid is never larger than the CU count, and every wait is
//          bounded (a timeout sets the status word in the workspace header and the sweep
//          finishes without waiting again).
//        * sgm_hwalk_kernel, left-to-right and right-to-left: one wave per image row.
//   2. TILES.  sgm_tile_kernel (D = 256, 64) / sgm_tile_h2_kernel (D = 128): one workgroup
//      per tile loads the tile's cost rows into LDS once, rebuilds all eight directions
//      inside the tile from the border states (every path crosses at most 8 pixels of a
//      tile), adds them in the reference order into an LDS-resident sum tile and writes the
//      finished rows (and/or their winner-takes-all).
//
// At D = 128 the interior waves of the sweeps and the tile kernel hold TWO pixels per wave
// (32 lanes x 4 disparities each, stm_sgm_step.h: sgm_step_h2*): the kernels are bound by
// vector instructions issued there, and a wave-wide minimum per 128 disparities is most of a
// step's cost.
//
// Traffic per (pixel, d): 5 x 4 (cost reads) + 2 x 6 (border states written and read)
// + 4 (sum written; 0 with the fused reduction) = 32-36 bytes.
//
// Same float operations in the same order as the reference -> identical bits (tested against
// the oracle and against the paired form).  Covers D = 64, 128, 256 (whole lanes, at most
// four values per lane: the tile's cost and sum must fit 160 KB of LDS); everything else
// takes the paired form.
#include "stm_sgm_tiled.h"

#include <stdlib.h>
#include <type_traits>

#include "stm_sgm_step.h"

namespace stm {
namespace {

typedef unsigned long long u64;

constexpr int kT = 8;             // tile side, = spacing of the stored border lines
constexpr int kSweepWaves = 16;   // waves per sweep workgroup (one workgroup per CU)
constexpr int kMboxSlots = 4;     // granule slots per mailbox (see mbox_* below)
constexpr unsigned kStatusTimeout = 1u;

// Geometry shared by the three kernels.  All offsets in floats from the workspace base.
struct TiledGeom {
  int H, W, D;
  int nTR, nTC;          // tile rows / columns
  long long rowb_off;    // [batch][6][nTR][W][D]: down V,A,B (rows y%8==7), up V,A,B (rows y%8==0)
  long long colb_off;    // [batch][6][nTC][H][D]: ->, dr, ur (columns x%8==7), <-, dl, ul (x%8==0)
  long long rowb_img, colb_img;
  long long mbox_off;    // [2 sweeps][batch][strips][2 sides][kMboxSlots][D] granules (u64)
  int strips;            // column strips per image in the sweeps
  int batch;
};

__device__ __forceinline__ float *rowb_ptr(float *ws, const TiledGeom &G, int b, int k, int t, int x) {
  return ws + G.rowb_off + (long long)b * G.rowb_img +
         (((long long)k * G.nTR + t) * G.W + x) * (long long)G.D;
}
__device__ __forceinline__ float *colb_ptr(float *ws, const TiledGeom &G, int b, int k, int tc, int y) {
  return ws + G.colb_off + (long long)b * G.colb_img +
         (((long long)k * G.nTC + tc) * G.H + y) * (long long)G.D;
}

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
  u64 *base = reinterpret_cast<u64 *>(ws + G.mbox_off);
  return base + ((((long long)sweep * G.batch + b) * G.strips + strip) * 2 + side) *
                    (long long)(kMboxSlots * G.D);
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
  u64 *slot = box + (size_t)(row % kMboxSlots) * D + d0;
  const unsigned tag = row + 1;
  if constexpr (VPL % 2 == 0) {
    const BufRsrc rs = make_rsrc(slot - d0, (unsigned)(D * sizeof(u64)));
#pragma unroll
    for (int j = 0; j < VPL; j += 2) {
      v4u_t q;
      q.x = __builtin_bit_cast(unsigned, r.v[j]); q.y = tag;
      q.z = __builtin_bit_cast(unsigned, r.v[j + 1]); q.w = tag;
      __builtin_amdgcn_raw_buffer_store_b128(q, rs, (d0 + j) * 8, 0, kAuxSc1);
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
  const u64 *slot = box + (size_t)(row % kMboxSlots) * D + d0;
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
    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
      dead = true;
      if ((threadIdx.x & 63) == 0) atomicOr(status, kStatusTimeout);
      return;
    }
  }
}

// a or b, element by element (a conditional on the structs would select between their
// addresses and put both in scratch memory).
template <int VPL>
__device__ __forceinline__ Row<VPL> pick_row(bool take_a, const Row<VPL> &a, const Row<VPL> &b) {
  Row<VPL> r;
#pragma unroll
  for (int j = 0; j < VPL; ++j) r.v[j] = take_a ? a.v[j] : b.v[j];
  return r;
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
#ifdef STM_TILE_STAMPS
  unsigned long long *stamps;  // diagnostic build only: [strip < 8][wave][row 500..515][8]
#endif
};

#ifdef STM_TILE_STAMPS
#define STM_SWEEP_STAMP(k)                                                                   \
  do {                                                                                       \
    if (P.stamps && !UP && blockIdx.y == 0 && s < 8 && t >= 100 && t < 116 && lane == 0) {   \
      unsigned long long t_;                                                                 \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
      P.stamps[(((size_t)s * NW + wv) * 16 + (t - 100)) * 8 + (k)] = t_;                     \
    }                                                                                        \
  } while (0)
#else
#define STM_SWEEP_STAMP(k) do {} while (0)
#endif

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
  const int xa = (s * NW + wv) * CW;  // this wave's first column
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

  // Element offsets (32-bit, wave-uniform) of this wave's columns in row 0; a wave at the right
  // border may own fewer than CW columns and re-reads its last one for the others.
  int xoff[CW];
#pragma unroll
  for (int j = 0; j < CW; ++j) xoff[j] = imin(xa + j, W - 1) * D;
  const int row_elems = W * D;
  auto row_of = [&](int t) { return UP ? H - 1 - t : t; };
  auto fetch = [&](int t, Row<VPL> (&dst)[CW]) {
    const int base = wave_uniform(row_of(imin(t, H - 1)) * row_elems);
#pragma unroll
    for (int j = 0; j < CW; ++j) dst[j] = load_row_at<VPL, true, true>(cost, (long)(unsigned)(base + xoff[j]), D, d0);
  };
  // Intensities for the 3*CW penalties of a row: lane l serves column l/3, direction l%3
  // (0 V, 1 A, 2 B) and reads its pixel and that pixel's predecessor on the path.
  const int pj = imin(lane / 3, CW - 1), pk = lane % 3;
  const int px = imin(xa + pj, W - 1);
  const int pxs = pk == 1 ? (px > 0 ? px - 1 : 0) : (pk == 2 ? imin(px + 1, W - 1) : px);
  auto fetch_intensity = [&](int t, float &icur, float &iprev) {
    const int tt = imin(t, H - 1);
    const int y = row_of(tt), yp = row_of(tt > 0 ? tt - 1 : 0);
    icur = image[(size_t)y * W + px];
    iprev = image[(size_t)yp * W + pxs];
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
      constexpr int PFH = 2;  // cost rows in flight beyond the current one
      // Addresses are a wave-uniform base plus this lane's 32-bit offset (its half's first
      // column and its disparities).
      const unsigned loff = (unsigned)(half * 2 * D + dh);
      const int col0 = xa + 2 * half;  // this half's columns: col0, col0 + 1
      const unsigned lload0 = (unsigned)((imin(col0, W - 1) - xa) * D + dh);
      const unsigned lload1 = (unsigned)((imin(col0 + 1, W - 1) - xa) * D + dh);
      const bool valid0 = col0 < W, valid1 = col0 + 1 < W;
      const bool start_a0 = col0 == 0;                             // a rightward path begins in image column 0
      const bool start_b0 = col0 >= W - 1, start_b1 = col0 + 1 >= W - 1;  // a leftward one in column W - 1
      const long row_step = (long)(UP ? -1 : 1) * W * D;
      const float *cnext = cost + (size_t)(UP ? H - 1 : 0) * W * D + (size_t)xa * D;  // next row to fetch
      int t_next = 0;
      // Penalties: lane i of every 16-lane row serves column i / 3 (0, 1) of the row's half and
      // direction i % 3 (0 V, 1 A, 2 B).
      const int i16 = lane & 15;
      const int pxh = imin(col0 + imin(i16 / 3, 1), W - 1), pkh = i16 % 3;
      const int pxsh = pkh == 1 ? (pxh > 0 ? pxh - 1 : 0) : (pkh == 2 ? imin(pxh + 1, W - 1) : pxh);
      const float *inext = image + (size_t)(UP ? H - 1 : 0) * W;
      const int istep = UP ? -W : W;
      Row<4> Vh[2], Ah[2], Bh[2], ch[PFH + 1][2];
      float ich[PFH + 1], iph[PFH + 1];
      auto refill = [&](Row<4> (&dst)[2], float &icur, float &iprev) {
        icur = inext[(unsigned)pxh];
        iprev = (t_next > 0 && H > 1 ? inext - istep : inext)[(unsigned)pxsh];
        dst[0] = load_row<4, true, true>(cnext + lload0, 0, D, 0);
        dst[1] = load_row<4, true, true>(cnext + lload1, 0, D, 0);
        if (t_next < H - 1) { cnext += row_step; inext += istep; }
        ++t_next;
      };
      const bool colA_here = (xa & (kT - 1)) == 4, colB_here = (xa & (kT - 1)) == 0;  // x % 8 == 7 is xa + 3, == 0 is xa
      const int y_first = UP ? H - 1 : 0;
      float *colA = colA_here && xa + 3 < W ? colb_ptr(ws, G, b, UP ? 2 : 1, (xa + 3) >> 3, y_first) : nullptr;
      float *colB = colB_here ? colb_ptr(ws, G, b, UP ? 5 : 4, xa >> 3, y_first) : nullptr;
      const int col_step = UP ? -D : D;
      float *rowV = rowb_ptr(ws, G, b, UP ? 3 : 0, 0, xa);
      const long long kstride = (long long)G.nTR * W * D;
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
      // A wave at a strip edge reads one mailbox: A from the left (wave 0) or B from the right
      // (last wave); both only in a one-wave strip, which the sweeps never have.
      const u64 *sub = !left_lds ? subA : (!right_lds ? subB : nullptr);
      auto leave = [&](int par, unsigned t) {  // what leaves the half: A of its last column, B of its first
        if (pubA && half == 1) mbox_publish<4>(pubA, D, dh, t, Ah[1]);
        if (pubB && half == 0) mbox_publish<4>(pubB, D, dh, t, Bh[0]);
        store_row<4, true>(xbase + par * kPar + my_off, 0, D, 0, Ah[1]);
        store_row<4, true>(xbase + par * kPar + kAB + my_off, 0, D, 0, Bh[0]);
      };

#pragma unroll
      for (int u = 0; u <= PFH; ++u) refill(ch[u], ich[u], iph[u]);
      int y = y_first;
      for (int t0 = 0; t0 < H; t0 += PFH + 1) {
#pragma unroll
        for (int u = 0; u <= PFH; ++u) {
          const int t = t0 + u;
          if (t < H) {
            const int par = t & 1;
            const float pen = sgm_penalty(ich[u], iph[u], P.p1, P.p2);
            if (t == 0) {
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
                if (sub) {
                  Row<4> r;
                  if (dead || !mbox_take<4>(qa4, (unsigned)(t - 1), r))
                    mbox_wait<4>(sub, D, dh, (unsigned)(t - 1), r, dead, status);
                  if (!left_lds) {  // wave 0 of a strip: its first half takes A from the strip on the left
                    if (half == 0) st[1] = r;
                  } else {          // last wave: its second half takes B from the strip on the right
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
            if (sub && !dead && t + 1 < H) mbox_peek<4>(sub, D, dh, (unsigned)t, qa4);
            if ((y & (kT - 1)) == (UP ? 0 : kT - 1)) {
              float *r0 = rowV + (long long)(y >> 3) * W * D;
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                if (!edge || (j == 0 ? valid0 : valid1)) {
                  store_row<4, true, true>(r0 + (loff + (unsigned)(j * D)), 0, D, 0, Vh[j]);
                  store_row<4, true, true>(r0 + kstride + (loff + (unsigned)(j * D)), 0, D, 0, Ah[j]);
                  store_row<4, true, true>(r0 + 2 * kstride + (loff + (unsigned)(j * D)), 0, D, 0, Bh[j]);
                }
              }
            }
            if (colA) {
              if (half == 1) store_row<4, true, true>(colA + (unsigned)dh, 0, D, 0, Ah[1]);
              colA += col_step;
            }
            if (colB) {
              if (half == 0) store_row<4, true, true>(colB + (unsigned)dh, 0, D, 0, Bh[0]);
              colB += col_step;
            }
            y += UP ? -1 : 1;
            refill(ch[u], ich[u], iph[u]);
            lds_barrier();
          }
        }
      }
      return;
    }
    // (a wave past the image only keeps the row barriers)
    for (int t = 0; t < H; ++t) lds_barrier();
    return;
  } else if (VPL <= 2 && xa >= 1 && xa + CW <= W - 1) {
    const long row_step = (long)(UP ? -1 : 1) * W * D;  // elements from one row to the next
    const float *cnext = cost + (size_t)(UP ? H - 1 : 0) * W * D + (size_t)xa * D + d0;  // this lane's part of the next row to fetch
    int t_next = 0;
    // Intensity samples of this lane's penalty (cur: pixel xa + pj of the row, prv: its
    // predecessor on the path, one row back).
    const float *inext = image + (size_t)(UP ? H - 1 : 0) * W;
    const int istep = UP ? -W : W;
    auto refill = [&](Row<VPL> (&dst)[CW], float &icur, float &iprev) {
      icur = inext[px];
      iprev = (t_next > 0 && H > 1 ? inext - istep : inext)[pxs];
#pragma unroll
      for (int j = 0; j < CW; ++j) dst[j] = load_row<VPL, true, true>(cnext + j * D, 0, D, 0);
      if (t_next < H - 1) { cnext += row_step; inext += istep; }  // past the end: the last row again (unused)
      ++t_next;
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
    const int col_step = UP ? -D : D;
    float *rowV = rowb_ptr(ws, G, b, UP ? 3 : 0, 0, xa) + d0;  // line 0; line t is t * W * D further
    const long long kstride = (long long)G.nTR * W * D;

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
          if ((y & (kT - 1)) == (UP ? 0 : kT - 1)) {
            float *r0 = rowV + (long long)(y >> 3) * W * D;
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
          if ((y & (kT - 1)) == (UP ? 0 : kT - 1)) {
#pragma unroll
            for (int j = 0; j < CW; ++j) {
              if (xa + j < W) {
                float *r0 = rowb_ptr(ws, G, b, UP ? 3 : 0, y >> 3, xa + j);
                const long long kstride = (long long)G.nTR * G.W * D;
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

// =====================================================================================
// Horizontal walks: one wave per image row and direction; the state is kept where the path
// leaves a tile (x % 8 == 7 going right, x % 8 == 0 going left).
// =====================================================================================
struct HwalkArgs {
  TiledGeom G;
  float p1, p2;
  int dlim;
  int only_left;  // D = 128: the tile kernel carries the left-to-right states along its tile rows
};

template <int VPL>
__global__ __launch_bounds__(256) void sgm_hwalk_kernel(const float *__restrict__ cost_all,
                                                        const float *__restrict__ image_all,
                                                        float *__restrict__ ws, const HwalkArgs P) {
  constexpr int PF = 8;
  constexpr int D = 64 * VPL;
  const TiledGeom &G = P.G;
  const int lane = threadIdx.x & 63;
  int wave = wave_uniform((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int H = G.H, W = G.W;
  if (wave >= (P.only_left ? 1 : 2) * H * G.batch) return;
  const bool left = P.only_left ? true : (wave & 1);  // 0: left-to-right, 1: right-to-left
  if (!P.only_left) wave >>= 1;
  const int b = wave / H, y = wave - b * H;
  const int d0 = lane * VPL;
  const float *cost = cost_all + ((size_t)b * H + y) * (size_t)W * D;
  const float *image = image_all + ((size_t)b * H + y) * (size_t)W;
  StepCtx X;
  X.init(P.p1, d0, P.dlim, lane == 63 && P.dlim < D);
  const int last = W - 1;
  auto col_of = [&](int i) { return left ? last - i : i; };

  // Penalties of 64 steps at a time (as in sgm_ckpt_kernel): lane l holds step i0 + l.
  float na = image[col_of(0)], nb, pen = 0.0f;
  auto pen_inputs = [&](int i0) {
    const float carry = read_lane(na, kWave - 1);
    na = image[col_of(imin(i0 + lane, last))];
    nb = lane_up(na, carry);
  };
  pen_inputs(0);

  Row<VPL> cbuf[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) cbuf[u] = load_row<VPL, true>(cost, (size_t)col_of(imin(u, last)), D, d0);
  Row<VPL> prev = cbuf[0];
  const int keep = left ? 0 : kT - 1;
  const int k = left ? 3 : 0;
  for (int base = 0; base < W; base += PF) {
    if ((base & (kWave - 1)) == 0) {
      pen = sgm_penalty(na, nb, P.p1, P.p2);
      pen_inputs(base + kWave);
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int i = base + u;
      if (i < W) {
        const Row<VPL> c = cbuf[u];
        cbuf[u] = load_row<VPL, true, true>(cost, (size_t)col_of(imin(i + PF, last)), D, d0);
        if (i > 0) prev = sgm_step<VPL, true>(prev, c, read_lane(pen, i & (kWave - 1)), X);
        const int x = col_of(i);
        if ((x & (kT - 1)) == keep) store_row<VPL, true, true>(colb_ptr(ws, G, b, k, x >> 3, y), 0, D, d0, prev);
      }
    }
  }
}

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
#ifdef STM_TILE_STAMPS
  unsigned long long *stamps;  // diagnostic build only: [tile][wave][8] s_memtime stamps
#endif
};

#ifdef STM_TILE_STAMPS
// Diagnostic build (tools/tile_stamps.py): where a tile's time goes.  Never in the product.
#define STM_STAMP(k)                                                                        \
  do {                                                                                      \
    if (P.stamps && b == 0 && tile < 4096 && lane == 0) {                                   \
      unsigned long long t_;                                                                \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
      P.stamps[((size_t)tile * 16 + w16) * 8 + (k)] = t_;                                   \
    }                                                                                       \
  } while (0)
#else
#define STM_STAMP(k) do {} while (0)
#endif

struct Seg {
  int ex, ey, n;  // first pixel in F order (tile coordinates) and pixel count
};
struct PhaseGeom {
  Seg a, b;
};

template <int PHASE>
__device__ __forceinline__ Seg tile_line(int k, int tw, int th) {
  Seg s;
  if (PHASE == 0) { s.ex = 0; s.ey = k; s.n = k < th ? tw : 0; }
  else if (PHASE == 1) { s.ex = k; s.ey = 0; s.n = k < tw ? th : 0; }
  else if (PHASE == 2) {  // x - y = k - 7, walked (+1, +1)
    const int c = k - (kT - 1);
    const int xlo = c > 0 ? c : 0, xhi = imin(tw - 1, th - 1 + c);
    s.ex = xlo; s.ey = xlo - c; s.n = xhi >= xlo ? xhi - xlo + 1 : 0;
  } else {                // x + y = k, walked (-1, +1)
    const int xlo = k - th + 1 > 0 ? k - th + 1 : 0, xhi = imin(tw - 1, k);
    s.ex = xhi; s.ey = k - xhi; s.n = xhi >= xlo ? xhi - xlo + 1 : 0;
  }
  if (s.n <= 0) { s.ex = 0; s.ey = 0; s.n = 0; }
  return s;
}

template <int PHASE>
__device__ __forceinline__ PhaseGeom tile_geom(int w, int tw, int th) {
  PhaseGeom g;
  g.a = tile_line<PHASE>(w, tw, th);
  if (PHASE >= 2 && w < kT - 1) g.b = tile_line<PHASE>(w + kT, tw, th);
  else { g.b.ex = 0; g.b.ey = 0; g.b.n = 0; }
  // Keep the slots dense: an empty first segment hands its place to the second.
  if (g.a.n == 0) { g.a = g.b; g.b.n = 0; }
  return g;
}

// An image's two blocks of border states; offsets inside a block fit 32 bits (make_plan).
struct BorderBase {
  const float *rowb, *colb;
};
__device__ __forceinline__ BorderBase border_base(float *ws, const TiledGeom &G, int b) {
  BorderBase B;
  B.rowb = ws + G.rowb_off + (long long)b * G.rowb_img;
  B.colb = ws + G.colb_off + (long long)b * G.colb_img;
  return B;
}
__device__ __forceinline__ const float *rowb_at(const BorderBase &B, const TiledGeom &G, int k, int t, int x) {
  return B.rowb + (unsigned)(((k * G.nTR + t) * G.W + x) * G.D);
}
__device__ __forceinline__ const float *colb_at(const BorderBase &B, const TiledGeom &G, int k, int tc, int y) {
  return B.colb + (unsigned)(((k * G.nTC + tc) * G.H + y) * G.D);
}

// State of direction `dir` after pixel (gx, gy), which lies on one of the stored lines.
__device__ __forceinline__ const float *border_ptr(const BorderBase &B, const TiledGeom &G, int dir, int gx,
                                                   int gy) {
  switch (dir) {
    case 0: return colb_at(B, G, 0, gx >> 3, gy);
    case 1: return colb_at(B, G, 3, gx >> 3, gy);
    case 2: return rowb_at(B, G, 0, gy >> 3, gx);
    case 3: return rowb_at(B, G, 3, gy >> 3, gx);
    case 4: return (gy & 7) == 7 ? rowb_at(B, G, 1, gy >> 3, gx) : colb_at(B, G, 1, gx >> 3, gy);
    case 5: return (gy & 7) == 0 ? rowb_at(B, G, 5, gy >> 3, gx) : colb_at(B, G, 5, gx >> 3, gy);
    case 6: return (gy & 7) == 7 ? rowb_at(B, G, 2, gy >> 3, gx) : colb_at(B, G, 4, gx >> 3, gy);
    default: return (gy & 7) == 0 ? rowb_at(B, G, 4, gy >> 3, gx) : colb_at(B, G, 2, gx >> 3, gy);
  }
}

template <int PHASE>
struct PhaseDir {
  static constexpr int dx = PHASE == 0 ? 1 : (PHASE == 1 ? 0 : (PHASE == 2 ? 1 : -1));
  static constexpr int dy = PHASE == 0 ? 0 : 1;
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
  constexpr int dx = PhaseDir<PHASE>::dx, dy = PhaseDir<PHASE>::dy;
  constexpr int D = 64 * VPL;
  E.start = 0;
#pragma unroll
  for (int i = 0; i < Entries<VPL, PHASE>::N / 2; ++i) {
    const Seg &s = i == 0 ? g.a : g.b;
    const int fx = x0 + s.ex - dx, fy = y0 + s.ey - dy;
    const int ix = x0 + s.ex + s.n * dx, iy = y0 + s.ey + s.n * dy;
    const bool fin = s.n > 0 && fx >= 0 && fx < G.W && fy >= 0 && fy < G.H;
    const bool iin = s.n > 0 && ix >= 0 && ix < G.W && iy >= 0 && iy < G.H;
    // A segment that does not exist or begins at the image border reads a valid dummy address:
    // the load stays unconditional (one definition of the registers), its value is not used.
    const float *fp = fin ? border_ptr(B, G, 2 * PHASE, fx, fy) : B.rowb;
    const float *ip = iin ? border_ptr(B, G, 2 * PHASE + 1, ix, iy) : B.rowb;
    E.e[2 * i] = load_row<VPL, true>(fp, 0, D, d0);
    E.e[2 * i + 1] = load_row<VPL, true>(ip, 0, D, d0);
    if (!fin && s.n > 0) E.start |= 1u << (2 * i);
    if (!iin && s.n > 0) E.start |= 2u << (2 * i);
  }
}

// How often the inverse pass of a diagonal is added (and, on tall images, how often before
// the forward pass), following the reference's stored path ends (stm_sgm.hip, sgm_decode).
struct Counts {
  int inv, pre;
};
template <int PHASE>
__device__ __forceinline__ Counts seg_counts(const Seg &s, int x0, int y0, const TiledGeom &G,
                                             int true_paths) {
  Counts c;
  c.inv = 1;
  c.pre = 0;
  if (PHASE < 2 || true_paths || s.n == 0) return c;
  const int e = G.W - G.H;
  const int gx = x0 + s.ex, gy = y0 + s.ey;
  if (PHASE == 2) {
    const int dgn = gx - gy;
    if (e > 0) {
      if (dgn >= 0 && dgn < e) c.inv = 0;
      else if (dgn == e) c.inv = 1 + e;
    } else if (e < -1) {
      if (dgn > e && dgn < 0) c.inv = 0;
      else if (dgn == e) c.pre = -e - 1;
    }
  } else if (e > 0) {
    const int sum = gx + gy;
    if (sum >= G.H && sum <= G.W - 1) c.inv = 0;
    else if (sum == G.H - 1) c.inv = 1 + e;
  }
  return c;
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
__device__ __forceinline__ void tile_wta_lds(const float *St, int wave, int lane, int tw, int th, int W, int x0,
                                             int y0, int32_t *__restrict__ disp) {
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
  if (r == 0 && px < tw && py < th) disp[(size_t)(y0 + py) * W + (x0 + px)] = k;
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
  const Counts ca = seg_counts<PHASE>(g.a, x0, y0, P.G, P.true_paths);
  const Counts cb = seg_counts<PHASE>(g.b, x0, y0, P.G, P.true_paths);
#pragma unroll
  for (int s = 0; s < kT; ++s) {
    if (s < nT) {
      const int pix = sl.pix[s];
      Row<VPL> acc;
      if (PHASE == 0) {
        if (P.accumulate) {
          acc = load_row<VPL, true>(out, (size_t)(y0 + (pix >> 3)) * P.G.W + (x0 + (pix & (kT - 1))), D, d0);
        } else {
#pragma unroll
          for (int j = 0; j < VPL; ++j) acc.v[j] = 0.0f;
        }
      } else {
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
        const size_t q = (size_t)(y0 + (pix >> 3)) * P.G.W + (x0 + (pix & (kT - 1)));
        store_row<VPL, true>(out, q, D, d0, acc);
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
                                              const Row<VPL> (&LI)[kT], bool inv_a, bool inv_b, float *St, int W, int x0, int y0,
                                              float *__restrict__ out, int32_t *__restrict__ disp, int lane) {
  constexpr int D = 64 * VPL;
  const int d0 = lane * VPL;
  Row<VPL> sn;
  if (PHASE != 0) sn = lds_srow<VPL>(St, sl.pix[0], d0);
#pragma unroll
  for (int s = 0; s < kT; ++s) {
    const int pix = sl.pix[s];
    Row<VPL> acc;
    if (PHASE == 0) {
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
      const size_t q = (size_t)(y0 + (pix >> 3)) * W + (x0 + (pix & (kT - 1)));
      store_row<VPL, true>(out, q, D, d0, acc);
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
                                          Between between) {
  Row<VPL> LF[kT], LI[kT];
  Slots<PHASE> sl;
  sl.init(g);
  // Fast forms: full tile, no path begins in it, every inverse pass counts once or not at all
  // (the reference skips the inverse diagonals of 17-44 % of a wide image's pixels, Q6: those
  // segments' I chains are not even computed); what is left for the generic forms are the
  // tiles on the image border and on the re-walked corner diagonals.
  bool fast = full && E.start == 0 && !(PHASE == 0 && P.accumulate);  // (accumulating: pair 0 starts from `out`, tile_add)
  bool inv_a = true, inv_b = true;
  if (PHASE >= 2 && fast) {
    const Counts ca = seg_counts<PHASE>(g.a, x0, y0, P.G, P.true_paths);
    const Counts cb = seg_counts<PHASE>(g.b, x0, y0, P.G, P.true_paths);
    fast = ca.inv <= 1 && ca.pre == 0 && cb.inv <= 1 && cb.pre == 0;
    inv_a = ca.inv != 0;
    inv_b = cb.inv != 0;
  }
  if (fast) {
    tile_chains_fast<VPL, PHASE>(sl, E, inv_a, inv_b, Ct, Ip, P.p1, P.p2, X, lane, LF, LI);
    between();
    tile_add_fast<VPL, PHASE, FINAL>(sl, LF, LI, inv_a, inv_b, St, P.G.W, x0, y0, out, disp, lane);
  } else {
    tile_chains<VPL, PHASE>(sl, E, Ct, Ip, P.p1, P.p2, X, lane, LF, LI);
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
// Position of a tile, advanced by gridDim.x tiles at a time without divisions.
struct TileWalk {
  int b, trow, tcol;     // image, tile row, tile column
  int drow, dcol;        // gridDim.x = drow * nTC + dcol
  __device__ __forceinline__ void init(int first, int stride, const TiledGeom &G) {
    const int per_image = G.nTR * G.nTC;
    b = first / per_image;
    const int t = first - b * per_image;
    trow = t / G.nTC;
    tcol = t - trow * G.nTC;
    drow = stride / G.nTC;
    dcol = stride - drow * G.nTC;
  }
  __device__ __forceinline__ void next(const TiledGeom &G) {
    tcol += dcol;
    trow += drow;
    if (tcol >= G.nTC) { tcol -= G.nTC; ++trow; }
    while (trow >= G.nTR) { trow -= G.nTR; ++b; }
  }
};

template <int VPL, int FINAL>
// (the second launch bound asks for two resident workgroups per CU where the LDS allows
// them, D <= 128: 8 waves per SIMD, i.e. at most 64 VGPRs)
__global__ __launch_bounds__(2 * kT * 64, VPL >= 4 ? 4 : 8) void sgm_tile_kernel(const float *__restrict__ cost_all,
                                                               const float *__restrict__ image_all,
                                                               float *__restrict__ out_all,
                                                               int32_t *__restrict__ disp_all,
                                                               float *__restrict__ ws, const TileArgs P) {
  constexpr int D = 64 * VPL;
  __shared__ __attribute__((aligned(16))) float Ct[kT * kT * D];
  __shared__ __attribute__((aligned(16))) float St[kT * kT * st_stride(D)];
  __shared__ float Ip[(kT + 2) * (kT + 2)];
  const TiledGeom &G = P.G;
  const int lane = threadIdx.x & 63;
  const int w16 = wave_uniform((int)(threadIdx.x >> 6));
  const int grp = w16 >> 3, w = w16 & (kT - 1);
  const int d0 = lane * VPL;
  const size_t npix = (size_t)G.H * G.W;
  const int ntiles = G.batch * G.nTR * G.nTC;  // < 2^31 (shape_ok)
  const int stride = (int)gridDim.x;
  StepCtx X;
  X.init(P.p1, d0, P.dlim, lane == 63 && P.dlim < D);
  TileWalk cur_t, next_t;  // the tile being computed / the one being prefetched
  cur_t.init((int)blockIdx.x, stride, G);
  next_t = cur_t;
  int id = (int)blockIdx.x;

  if (grp == 0) {
    Row<VPL> cpre[kT];   // next tile: tile row w of the cost tile
    float ipre = 0.0f;   // next tile: this thread's element of the intensity patch
    Entries<VPL, 0> E0;  // next tile: border states of pair 0
    // Every prefetch redefines all of its registers unconditionally (past the last tile it
    // re-reads that tile, rows and samples outside a partial tile are clamped into it): a
    // conditional load would keep the previous contents alive around the whole loop.
    auto prefetch = [&](const TileWalk &t) {
      const int x0 = t.tcol * kT, y0 = t.trow * kT;
      const int tw = imin(kT, G.W - x0), th = imin(kT, G.H - y0);
      const float *cost = cost_all + (size_t)t.b * npix * D;
      const unsigned rowbase = (unsigned)((y0 + imin(w, th - 1)) * G.W + x0);
#pragma unroll
      for (int c = 0; c < kT; ++c)
        cpre[c] = load_row_at<VPL, true>(cost, (long)((rowbase + (unsigned)imin(c, tw - 1)) * (unsigned)D), D, d0);
      // Intensities of the tile and the ring of pixels around it (clamped at the image border;
      // a clamped sample is only ever used for a step that does not exist).
      {
        const int e = imin((int)threadIdx.x, (kT + 2) * (kT + 2) - 1);
        const int r = e / (kT + 2) - 1, c = e % (kT + 2) - 1;
        const int gy = imin(y0 + r > 0 ? y0 + r : 0, G.H - 1), gx = imin(x0 + c > 0 ? x0 + c : 0, G.W - 1);
        ipre = image_all[(size_t)t.b * npix + (unsigned)(gy * G.W + gx)];
      }
      load_entries<VPL, 0>(E0, tile_geom<0>(w, tw, th), border_base(ws, G, t.b), G, x0, y0, d0);
    };
    // The prefetched cost rows and intensities of tile `t` go to LDS (nobody reads Ct or Ip
    // between a tile's B3 and the next tile's B0).
    auto stage = [&](const TileWalk &t) {
      const int th = imin(kT, G.H - t.trow * kT);
      if (w < th) {
#pragma unroll
        for (int c = 0; c < kT; ++c) store_row<VPL, true>(Ct + (size_t)(w * kT + c) * D, 0, D, d0, cpre[c]);
      }
      if (threadIdx.x < (kT + 2) * (kT + 2)) Ip[threadIdx.x] = ipre;
    };
    prefetch(next_t);
    stage(next_t);
    for (; id < ntiles; id += stride) {
      const int b = cur_t.b, x0 = cur_t.tcol * kT, y0 = cur_t.trow * kT;
      const int tw = imin(kT, G.W - x0), th = imin(kT, G.H - y0);
      const bool full = tw == kT && th == kT;
      float *out = out_all ? out_all + (size_t)b * npix * D : nullptr;
      int32_t *disp = disp_all ? disp_all + (size_t)b * npix : nullptr;
#ifdef STM_TILE_STAMPS
      const int tile = id < 4096 ? id : 4096;
#endif
      STM_STAMP(0);
      const Entries<VPL, 0> E0now = E0;
      STM_STAMP(1);
      lds_barrier();  // B0
      STM_STAMP(2);
      tile_pair<VPL, 0, FINAL>(tile_geom<0>(w, tw, th), E0now, full, Ct, St, Ip, P, x0, y0, out, disp, X, lane,
                               [] {});
      STM_STAMP(3);
      const PhaseGeom g2 = tile_geom<2>(w, tw, th);
      Entries<VPL, 2> E2;
      load_entries<VPL, 2>(E2, g2, border_base(ws, G, b), G, x0, y0, d0);
      lds_barrier();  // B1
      STM_STAMP(4);
      tile_pair<VPL, 2, FINAL>(g2, E2, full, Ct, St, Ip, P, x0, y0, out, disp, X, lane, [&] {
        STM_STAMP(5);
        lds_barrier();  // B2
      });
      STM_STAMP(6);
      if (id + stride < ntiles) next_t.next(G);
      prefetch(next_t);
      lds_barrier();  // B3
      stage(next_t);  // (while group 1 runs its last update)
      if constexpr (FINAL != 0) {
        lds_barrier();  // B4: the finished sum tile is complete
        tile_wta_lds<D, 2 * kT>(St, w16, lane, tw, th, G.W, x0, y0, disp);
      }
      cur_t = next_t;
      STM_STAMP(7);
    }
  } else {
    Entries<VPL, 1> E1;  // next tile: border states of pair 1
    auto prefetch = [&](const TileWalk &t) {
      const int x0 = t.tcol * kT, y0 = t.trow * kT;
      const int tw = imin(kT, G.W - x0), th = imin(kT, G.H - y0);
      load_entries<VPL, 1>(E1, tile_geom<1>(w, tw, th), border_base(ws, G, t.b), G, x0, y0, d0);
    };
    prefetch(next_t);
    for (; id < ntiles; id += stride) {
      const int b = cur_t.b, x0 = cur_t.tcol * kT, y0 = cur_t.trow * kT;
      const int tw = imin(kT, G.W - x0), th = imin(kT, G.H - y0);
      const bool full = tw == kT && th == kT;
      float *out = out_all ? out_all + (size_t)b * npix * D : nullptr;
      int32_t *disp = disp_all ? disp_all + (size_t)b * npix : nullptr;
#ifdef STM_TILE_STAMPS
      const int tile = id < 4096 ? id : 4096;
#endif
      STM_STAMP(0);
      const Entries<VPL, 1> E1now = E1;
      STM_STAMP(1);
      lds_barrier();  // B0
      STM_STAMP(2);
      tile_pair<VPL, 1, FINAL>(tile_geom<1>(w, tw, th), E1now, full, Ct, St, Ip, P, x0, y0, out, disp, X, lane,
                               [&] {
                                 STM_STAMP(3);
                                 lds_barrier();  // B1
                               });
      const PhaseGeom g3 = tile_geom<3>(w, tw, th);
      Entries<VPL, 3> E3;
      load_entries<VPL, 3>(E3, g3, border_base(ws, G, b), G, x0, y0, d0);
      STM_STAMP(4);
      lds_barrier();  // B2
      if (id + stride < ntiles) next_t.next(G);
      tile_pair<VPL, 3, FINAL>(g3, E3, full, Ct, St, Ip, P, x0, y0, out, disp, X, lane, [&] {
        prefetch(next_t);
        STM_STAMP(5);
        lds_barrier();  // B3
        STM_STAMP(6);
      });
      if constexpr (FINAL != 0) {
        lds_barrier();  // B4
        tile_wta_lds<D, 2 * kT>(St, w16, lane, tw, th, G.W, x0, y0, disp);
      }
      cur_t = next_t;
      STM_STAMP(7);
    }
  }
}

// ---- D = 128: two lines of a pair per wave ---------------------------------------------------
// sgm_tile_h2_kernel: four waves own a tile (cost + sum tile: 64 KB of LDS, two workgroups per
// CU) and take it through the four pairs; each wave steps two lines of a pair at once, one
// per half (sgm_step_h2).  On the diagonal pairs a line is two segments (k+1 and 7-k pixels);
// a wave's two lines are chosen so that both change segment at the same slot -- line w in the
// order (first, second) with line 6-w in the order (second, first), w = 0..2; wave 3 takes
// line 3 (4 + 4) and the main diagonal (8, cut into 4 + 4 that continue) -- which keeps the
// control flow wave-uniform.  Interior tiles whose inverse passes all count once (or all not
// at all) take this form; the others (image border, the reference's re-walked diagonals: a
// few per cent) run the one-line-per-wave forms above, two lines after each other per wave.
struct H2Part {
  int ex, ey, n;
};

template <int PHASE>
__device__ __forceinline__ void h2_parts(int w, int half, H2Part &p0, H2Part &p1, bool &restart) {
  constexpr int dx = PhaseDir<PHASE>::dx, dy = PhaseDir<PHASE>::dy;
  auto part = [](const Seg &s) { H2Part p; p.ex = s.ex; p.ey = s.ey; p.n = s.n; return p; };
  if (PHASE < 2) {
    p0 = part(tile_line<PHASE>(2 * w + half, kT, kT));
    p1.ex = p1.ey = p1.n = 0;
    restart = false;
  } else if (w < 3) {
    const int j = half ? 6 - w : w;
    p0 = part(tile_line<PHASE>(half ? j + kT : j, kT, kT));
    p1 = part(tile_line<PHASE>(half ? j : j + kT, kT, kT));
    restart = true;
  } else if (!half) {
    p0 = part(tile_line<PHASE>(3, kT, kT));
    p1 = part(tile_line<PHASE>(3 + kT, kT, kT));
    restart = true;
  } else {
    p0 = part(tile_line<PHASE>(kT - 1, kT, kT));
    p0.n = 4;
    p1.ex = p0.ex + 4 * dx; p1.ey = p0.ey + 4 * dy; p1.n = 4;
    restart = false;
  }
}

__device__ __forceinline__ constexpr int h2_na(int phase, int w) { return phase < 2 ? kT : (w < 3 ? w + 1 : 4); }

// Offset (floats) of the state of direction `dir` after the pixel (rx, ry) relative to a
// tile's origin, rx, ry in [-1, 8], inside an image's block of row (is_row) or column border
// states, minus the tile's own term ((trow * W + x0) * D resp. (tcol * H + y0) * D); see
// border_ptr.
__device__ __forceinline__ int h2_entry_offset(const TiledGeom &G, int dir, int rx, int ry, bool &is_row) {
  int k;
  switch (dir) {
    case 0: is_row = false; k = 0; break;
    case 1: is_row = false; k = 3; break;
    case 2: is_row = true; k = 0; break;
    case 3: is_row = true; k = 3; break;
    case 4: is_row = (ry & 7) == 7; k = 1; break;
    case 5: is_row = (ry & 7) == 0; k = 5; break;
    case 6: is_row = (ry & 7) == 7; k = is_row ? 2 : 4; break;
    default: is_row = (ry & 7) == 0; k = is_row ? 4 : 2; break;
  }
  if (is_row) return ((k * G.nTR + (ry >> 3)) * G.W + rx) * G.D;
  return ((k * G.nTC + (rx >> 3)) * G.H + ry) * G.D;
}

// Per-lane constants of a pair: where the lane's line lies in the tile and where its four
// entry states lie relative to a tile.
struct H2Pair {
  int pix0, pix1;   // tile pixel index (y * 8 + x) of the first slot of either part
  bool restart;     // the second part begins at an entry state (it is another diagonal)
  int eoff[4];      // F / I entry of the first part, F / I entry of the second: offset in floats (a
                    // multiple of D) | 1 when it lies in the row border states
};

template <int PHASE>
__device__ __forceinline__ void h2_pair_init(H2Pair &K, const TiledGeom &G, int w, int half) {
  constexpr int dx = PhaseDir<PHASE>::dx, dy = PhaseDir<PHASE>::dy;
  H2Part p[2];
  h2_parts<PHASE>(w, half, p[0], p[1], K.restart);
  K.pix0 = p[0].ey * kT + p[0].ex;
  K.pix1 = p[1].ey * kT + p[1].ex;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    bool rf, ri;
    const int of = h2_entry_offset(G, 2 * PHASE, p[i].ex - dx, p[i].ey - dy, rf);
    const int oi = h2_entry_offset(G, 2 * PHASE + 1, p[i].ex + p[i].n * dx, p[i].ey + p[i].n * dy, ri);
    K.eoff[2 * i] = of | (rf ? 1 : 0);
    K.eoff[2 * i + 1] = oi | (ri ? 1 : 0);
  }
}

// Position of a tile for a workgroup that owns whole tile rows (rows blockIdx.x, + gridDim.x,
// ... of batch * nTR) and walks each from left to right.
struct TileRowWalk {
  int b, trow, tcol, row;
  __device__ __forceinline__ void init(int r, const TiledGeom &G) {
    row = r;
    b = r / G.nTR;
    trow = r - b * G.nTR;
    tcol = 0;
  }
  __device__ __forceinline__ void next(const TiledGeom &G, int stride) {
    if (++tcol == G.nTC) init(row + stride, G);
  }
};

// A tile's terms of the entry addresses; real = false: the tile is not interior, its entry
// states are not used and every lane reads the same valid dummy row instead (`row`).
struct H2TileBase {
  const float *row, *col;
  bool real;
};
template <typename Walk>
__device__ __forceinline__ H2TileBase h2_tile_base(float *ws, const TiledGeom &G, const Walk &t) {
  H2TileBase T;
  T.row = ws + G.rowb_off + (long long)t.b * G.rowb_img + (long long)(t.trow * G.W + t.tcol * kT) * G.D;
  T.col = ws + G.colb_off + (long long)t.b * G.colb_img + (long long)(t.tcol * G.H + t.trow * kT) * G.D;
  T.real = true;
  return T;
}

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

// How the inverse passes of a pair count inside a tile (seg_counts): 1 = every one once, 0 =
// none at all (the reference never runs them there), 2 = some once and some not (per line),
// 3 = the tile holds the diagonal that counts several times or before its forward pass (the
// update then repeats its additions per line).
template <int PHASE>
__device__ __forceinline__ int h2_count_class(const TiledGeom &G, int x0, int y0, int true_paths) {
  if (PHASE < 2 || true_paths) return 1;
  const int e = G.W - G.H;
  if (PHASE == 2) {
    const int lo = x0 - y0 - (kT - 1), hi = x0 - y0 + (kT - 1);  // x - y over the tile
    if (e > 0) {
      if (lo <= e && e <= hi) return 3;
      if (lo >= 0 && hi <= e - 1) return 0;
      return (hi < 0 || lo > e) ? 1 : 2;
    }
    if (e < -1) {
      if (lo <= e && e <= hi) return 3;
      if (lo >= e + 1 && hi <= -1) return 0;
      return (hi < e || lo >= 0) ? 1 : 2;
    }
    return 1;
  }
  if (e > 0) {
    const int lo = x0 + y0, hi = x0 + y0 + 2 * (kT - 1);  // x + y over the tile
    if (lo <= G.H - 1 && G.H - 1 <= hi) return 3;
    if (lo >= G.H && hi <= G.W - 1) return 0;
    return (hi < G.H - 1 || lo > G.W - 1) ? 1 : 2;
  }
  return 1;
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

__device__ __forceinline__ Row<4> h2_lds_row(const float *tile, int byte_addr) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f q = *reinterpret_cast<const v4f *>(reinterpret_cast<const char *>(tile) + byte_addr);
  Row<4> r;
  r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w;
  return r;
}
__device__ __forceinline__ void h2_lds_store(float *tile, int byte_addr, const Row<4> &r) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f q;
  q.x = r.v[0]; q.y = r.v[1]; q.z = r.v[2]; q.w = r.v[3];
  *reinterpret_cast<v4f *>(reinterpret_cast<char *>(tile) + byte_addr) = q;
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
    h2_parts<PHASE>(w, half, q0, q1, restart_);
    Seg s0, s1;
    s0.ex = q0.ex; s0.ey = q0.ey; s0.n = q0.n;
    s1.ex = q1.ex; s1.ey = q1.ey; s1.n = q1.n;
    const Counts c0 = seg_counts<PHASE>(s0, x0, y0, G, true_paths), c1 = seg_counts<PHASE>(s1, x0, y0, G, true_paths);
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
    if (PHASE == 0) {
      if (accumulate) {  // STM_SGM_ACCUMULATE: the sum starts from the caller's contents of `out`
        const int p = s < NA ? K.pix0 + s * pstep : K.pix1 + (s - NA) * pstep;
        acc = load_row<4, true>(out, (size_t)((y0 + (p >> 3)) * W + (x0 + (p & (kT - 1)))), 128, hl16 >> 2);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc.v[j] = 0.0f;
      }
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
      const unsigned qa = (unsigned)((y0 + (pa >> 3)) * W + (x0 + (pa & (kT - 1))));  // < H * W
      const unsigned qb = (unsigned)((y0 + (pb >> 3)) * W + (x0 + (pb & (kT - 1))));
      store_row<4, true>(out, (size_t)qa, 128, d0, ra);
      store_row<4, true>(out, (size_t)qb, 128, d0, rb);
    }
  };
  auto sum_of = [&](auto s_) {
    Row<4> r;
    if (PHASE != 0) r = h2_lds_row(St, saddr(s_));
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

// A pair of a tile that does not qualify, for one wave: its two lines after each other in the
// one-line-per-wave forms (two values per lane).  The entry states of all four pairs are
// requested at the start of the tile (h2_generic_entries): a global load takes longer than a
// pair, and eight of them in a row were most of such a tile.
template <int PHASE>
struct H2GenericEntries {
  PhaseGeom g[2];
  Entries<2, PHASE> E[2];
};
template <int PHASE>
__device__ __forceinline__ void h2_generic_entries(H2GenericEntries<PHASE> &Q, int w, int tw, int th,
                                                   const TileArgs &P, const BorderBase &B, int x0, int y0, int lane) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    Q.g[i] = tile_geom<PHASE>(2 * w + i, tw, th);
    load_entries<2, PHASE>(Q.E[i], Q.g[i], B, P.G, x0, y0, lane * 2);
  }
}

template <int PHASE, int FINAL, typename Between>
__device__ __forceinline__ void h2_pair_generic(const H2GenericEntries<PHASE> &Q, int w, int tw, int th,
                                                const float *Ct, float *St, const float *Ip, const TileArgs &P,
                                                int x0, int y0, float *__restrict__ out,
                                                int32_t *__restrict__ disp, StepCtx &X, int lane, float *exits,
                                                Between between) {
  const int d0 = lane * 2;
  Slots<PHASE> sl[2];
  Row<2> LF[2][kT], LI[2][kT];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    sl[i].init(Q.g[i]);
    tile_chains<2, PHASE>(sl[i], Q.E[i], Ct, Ip, P.p1, P.p2, X, lane, LF[i], LI[i]);
    if (PHASE == 0 && exits && tw == kT && 2 * w + i < th)  // the rightward state where row 2 w + i leaves the tile
      store_row<2, true>(exits + (size_t)(2 * w + i) * 128, 0, 128, d0, LF[i][kT - 1]);
  }
  between();
#pragma unroll
  for (int i = 0; i < 2; ++i)
    tile_add<2, PHASE, FINAL>(Q.g[i], sl[i], LF[i], LI[i], St, P, x0, y0, out, disp, lane);
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
  H2Pair K0, K1, K2, K3;
  h2_pair_init<0>(K0, G, w, half);
  h2_pair_init<1>(K1, G, w, half);
  h2_pair_init<2>(K2, G, w, half);
  h2_pair_init<3>(K3, G, w, half);
  // A workgroup owns whole tile rows and walks them left to right: the rightward state with
  // which a row leaves a tile is the next tile's entry state and stays in registers (between
  // two tiles of this form; otherwise it goes through the column border states, which every
  // tile writes) -- there is no left-to-right horizontal walk at D = 128, i.e. one read of the
  // cost volume fewer.
  const int nrows = G.batch * G.nTR;
  TileRowWalk cur_t, next_t;  // the tile being computed / the one being prefetched
  cur_t.init((int)blockIdx.x, G);
  if (cur_t.row >= nrows) return;
  next_t = cur_t;

  // A tile is interior when all of its neighbours exist (every line enters from a border state).
  auto interior = [&](const TileRowWalk &t) {
    return t.trow >= 1 && t.trow <= G.nTR - 2 && t.tcol >= 1 && t.tcol <= G.nTC - 2 &&
           (t.trow + 1) * kT <= G.H && (t.tcol + 1) * kT <= G.W;
  };
  Row<4> cpre[kT];  // next tile: tile row 2 w + half of the cost tile
  float ipre = 0.0f;
  Row<4> E0[4], E1[4];  // next tile: entry states of pairs 0 and 1 (interior tiles)
  Row<4> carry;         // the previous tile's rightward exit state of row 2 w + half
  bool have_carry = false;
  // Every prefetch redefines all of its registers unconditionally (past the last tile it
  // re-reads that tile; rows and samples outside a partial tile are clamped into it; the
  // entry states of a tile that is not interior are read at the workspace's first border row).
  auto prefetch_cost = [&](const TileRowWalk &t) {
    const int x0 = t.tcol * kT, y0 = t.trow * kT;
    const int tw = imin(kT, G.W - x0), th = imin(kT, G.H - y0);
    const float *cost = cost_all + (size_t)t.b * npix * D;
    const unsigned rowbase = (unsigned)((y0 + imin(2 * w + half, th - 1)) * G.W + x0);
#pragma unroll
    for (int c = 0; c < kT; ++c)
      cpre[c] = load_row_at<4, true>(cost, (long)((rowbase + (unsigned)imin(c, tw - 1)) * (unsigned)D), D, d0);
    const int e = imin((int)threadIdx.x, (kT + 2) * (kT + 2) - 1);
    const int r = e / (kT + 2) - 1, c = e % (kT + 2) - 1;
    const int gy = imin(y0 + r > 0 ? y0 + r : 0, G.H - 1), gx = imin(x0 + c > 0 ? x0 + c : 0, G.W - 1);
    ipre = image_all[(size_t)t.b * npix + (unsigned)(gy * G.W + gx)];
  };
  auto entry_base = [&](const TileRowWalk &t) {
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
#ifdef STM_TILE_STAMPS
    const int tid_ = cur_t.row * G.nTC + cur_t.tcol;
    const int tile = tid_ < 4096 ? tid_ : 4096;
    const int w16 = w;
#endif
    STM_STAMP(0);
    const int cls2 = h2_count_class<2>(G, x0, y0, P.true_paths), cls3 = h2_count_class<3>(G, x0, y0, P.true_paths);
    const bool fast = interior(cur_t) && cls2 >= 0 && cls3 >= 0;
    if (2 * w + half < th) {
#pragma unroll
      for (int c = 0; c < kT; ++c) h2_lds_store(Ct, ((2 * w + half) * kT + c) * D * 4 + hl16, cpre[c]);
    }
    if (threadIdx.x < (kT + 2) * (kT + 2)) Ip[threadIdx.x] = ipre;
    Row<4> Ea[4], Eb[4];
    Ea[0] = pick_row(have_carry, carry, E0[0]);
    Ea[1] = E0[1];
    Eb[0] = E1[0]; Eb[1] = E1[1];
    STM_STAMP(1);
    lds_barrier();
    STM_STAMP(2);
    bool more;
    {
      TileRowWalk n = cur_t;
      n.next(G, stride);
      more = n.row < nrows;
      if (more) next_t = n;
    }
    // Where the rows of this tile leave it to the right (column border states of direction 0).
    float *exits = colb_ptr(ws, G, b, 0, cur_t.tcol, y0);
    // A global load takes several times as long as a pair (2-4 us against ~1 us), and results
    // return in issue order: the entry states of pairs 2 and 3 are requested now, then the next
    // tile's cost rows; the next tile's pairs 0 and 1 follow once pairs 0 and 1 are done.
    if (fast) {
      const H2TileBase T = h2_tile_base(ws, G, cur_t);
      Row<4> Ec[4], Ed[4], unused;
      h2_load_entries<4>(Ec, K2, T, d0);
      h2_load_entries<4>(Ed, K3, T, d0);
      prefetch_cost(next_t);
      // The penalties of all four pairs now: four independent division sequences instead of one
      // in front of every pair's chains.
      const int na = h2_na(2, w);
      const float pen0 = h2_penalties<0>(K0, kT, Ip, P.p1, P.p2, lane);
      const float pen1 = h2_penalties<1>(K1, kT, Ip, P.p1, P.p2, lane);
      const float pen2 = h2_penalties<2>(K2, na, Ip, P.p1, P.p2, lane);
      const float pen3 = h2_penalties<3>(K3, na, Ip, P.p1, P.p2, lane);
      h2_pair_w<0, FINAL>(w, half, K0, Ea, 1, P, Ct, St, pen0, X, lane, hl16, x0, y0, out, disp, carry, [] {});
      have_carry = true;
      store_row<4, true>(exits + (size_t)(2 * w + half) * D, 0, D, d0, carry);
      h2_pair_w<1, FINAL>(w, half, K1, Eb, 1, P, Ct, St, pen1, X, lane, hl16, x0, y0, out, disp, unused,
                          [] { lds_barrier(); });
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
      h2_pair_generic<0, FINAL>(Q0, w, tw, th, Ct, St, Ip, P, x0, y0, out, disp, X1, lane, exits, [] {});
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
    if constexpr (FINAL != 0) tile_wta_lds<D, 4>(St, w, lane, tw, th, G.W, x0, y0, disp);
    if (!more) break;
    cur_t = next_t;
  }
}

// ---- host side ------------------------------------------------------------------------
#ifdef STM_TILE_STAMPS
unsigned long long *g_tile_stamps = nullptr;
unsigned long long *g_sweep_stamps = nullptr;
#endif
int sweep_cw(int VPL) { return 8 / VPL; }  // columns per wave: 24 state registers

int device_cus() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  return n;
}

struct TiledPlan {
  TiledGeom G;
  long long status_off;
  size_t total_floats;
  size_t zero_bytes;  // status word + mailboxes, cleared at the start of every call
  int group;          // images per sweep launch
};

bool make_plan(int H, int W, int D, int batch, TiledPlan &plan) {
  const int VPL = D / 64;
  const int cus = device_cus();
  if (cus <= 0) return false;
  const int strip_cols = kSweepWaves * sweep_cw(VPL);
  const int strips = (W + strip_cols - 1) / strip_cols;
  if (strips > cus) return false;
  TiledGeom &G = plan.G;
  G.H = H; G.W = W; G.D = D;
  G.nTR = (H + kT - 1) / kT;
  G.nTC = (W + kT - 1) / kT;
  G.strips = strips;
  G.batch = batch;
  plan.group = cus / strips;
  if (plan.group > batch) plan.group = batch;
  long long off = 0;
  plan.status_off = off;
  off += 64;  // 256 bytes: status word, padding to keep what follows 16-byte aligned
  G.mbox_off = off;
  off += 2ll * batch * strips * 2 * kMboxSlots * D * 2;  // granules are two floats wide
  plan.zero_bytes = (size_t)off * sizeof(float);
  // Buffer descriptors address one image's cost volume / border-state block with 32-bit
  // byte offsets.
  if ((long long)H * W * D * 4 >= (1ll << 32) || 6ll * G.nTR * W * D * 4 >= (1ll << 32) ||
      6ll * G.nTC * H * D * 4 >= (1ll << 32))
    return false;
  G.rowb_off = off;
  G.rowb_img = 6ll * G.nTR * W * D;
  off += G.rowb_img * batch;
  G.colb_off = off;
  G.colb_img = 6ll * G.nTC * H * D;
  off += G.colb_img * batch;
  plan.total_floats = (size_t)off;
  return true;
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
bool sweep_fits(int D) {
  static signed char cache[16][3];  // [device][VPL index]: 0 unknown, 1 fits, -1 does not
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  const int k = D == 64 ? 0 : (D == 128 ? 1 : 2);
  if (dev >= 0 && dev < 16 && cache[dev][k] != 0) return cache[dev][k] > 0;
  const bool ok = D == 64 ? sweep_fits_vpl<1>() : (D == 128 ? sweep_fits_vpl<2>() : sweep_fits_vpl<4>());
  if (dev >= 0 && dev < 16) cache[dev][k] = ok ? 1 : -1;
  return ok;
}

template <int VPL>
int run_tiled(const float *cost, const float *left, float *out, int32_t *disp, const TiledPlan &plan,
              float p1, float p2, int dlim, int true_paths, int accumulate, bool cooperative, float *ws, int batch,
              hipStream_t s, int final_mode) {
  const TiledGeom &G = plan.G;
  if (hipMemsetAsync(ws, 0, plan.zero_bytes, s) != hipSuccess)
    return fail(STM_EHIP, "stm_sgm_f32 (tiled): hipMemsetAsync failed");
#ifdef STM_TILE_STAMPS
  // diagnostic build only: leave out the walks (1), the sweeps (2), the tiles (4)
  const int skip = getenv("STM_TILED_SKIP") ? atoi(getenv("STM_TILED_SKIP")) : 0;
#else
  constexpr int skip = 0;
#endif
  if (!(skip & 1)) {
    HwalkArgs A;
    A.G = G; A.p1 = p1; A.p2 = p2; A.dlim = dlim;
    A.only_left = VPL == 2;
    const long waves = (A.only_left ? 1l : 2l) * G.H * batch;
    hipLaunchKernelGGL((sgm_hwalk_kernel<VPL>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s,
                       cost, left, ws, A);
    int rc = check_launch("stm_sgm_f32 (tiled: horizontal walks)");
    if (rc != STM_OK) return rc;
  }
  for (int up = 0; up < 2 && !(skip & 2); ++up) {
    for (int b0 = 0; b0 < batch; b0 += plan.group) {
      SweepArgs A;
      A.G = G; A.p1 = p1; A.p2 = p2; A.dlim = dlim; A.b0 = b0; A.status_off = plan.status_off;
#ifdef STM_TILE_STAMPS
      A.stamps = g_sweep_stamps;
#endif
      const int n = batch - b0 < plan.group ? batch - b0 : plan.group;
      const dim3 grid(G.strips, n), block(kSweepWaves * 64);
      constexpr int CW = 8 / VPL;  // = sweep_cw(VPL)
      const int rc = up ? launch_sweep(sgm_sweep_kernel<VPL, CW, true>, grid, block, s, cooperative, cost, left, ws, A)
                        : launch_sweep(sgm_sweep_kernel<VPL, CW, false>, grid, block, s, cooperative, cost, left, ws, A);
      if (rc != STM_OK) return rc;
    }
  }
  if (skip & 4) return STM_OK;
  {
    TileArgs A;
    A.G = G; A.p1 = p1; A.p2 = p2; A.dlim = dlim; A.true_paths = true_paths; A.accumulate = accumulate;
#ifdef STM_TILE_STAMPS
    A.stamps = g_tile_stamps;
#endif
    // Persistent workgroups; tiles are dealt round robin.
    const long long ntiles = (long long)G.nTR * G.nTC * batch;
    if constexpr (VPL == 2) {
      // D = 128: four-wave workgroups, two lines of a pair per wave, two workgroups per CU
      const long long wgs = (long long)device_cus() * 2, rows = (long long)G.nTR * batch;
      const dim3 grid((unsigned)(rows < wgs ? rows : wgs)), block(4 * 64);
      if (final_mode == 0)
        hipLaunchKernelGGL((sgm_tile_h2_kernel<0>), grid, block, 0, s, cost, left, out, disp, ws, A);
      else if (final_mode == 1)
        hipLaunchKernelGGL((sgm_tile_h2_kernel<1>), grid, block, 0, s, cost, left, out, disp, ws, A);
      else
        hipLaunchKernelGGL((sgm_tile_h2_kernel<2>), grid, block, 0, s, cost, left, out, disp, ws, A);
    } else if constexpr (VPL >= 4) {
      // D = 256: one sixteen-wave workgroup per CU (the LDS allows no more)
      const long long wgs = device_cus();
      const dim3 grid((unsigned)(ntiles < wgs ? ntiles : wgs)), block(2 * kT * 64);
      if (final_mode == 0)
        hipLaunchKernelGGL((sgm_tile_kernel<VPL, 0>), grid, block, 0, s, cost, left, out, disp, ws, A);
      else if (final_mode == 1)
        hipLaunchKernelGGL((sgm_tile_kernel<VPL, 1>), grid, block, 0, s, cost, left, out, disp, ws, A);
      else
        hipLaunchKernelGGL((sgm_tile_kernel<VPL, 2>), grid, block, 0, s, cost, left, out, disp, ws, A);
    } else {
      // D = 64: sixteen-wave workgroups, two per CU
      const long long wgs = (long long)device_cus() * 2;
      const dim3 grid((unsigned)(ntiles < wgs ? ntiles : wgs)), block(2 * kT * 64);
      if (final_mode == 0)
        hipLaunchKernelGGL((sgm_tile_kernel<VPL, 0>), grid, block, 0, s, cost, left, out, disp, ws, A);
      else if (final_mode == 1)
        hipLaunchKernelGGL((sgm_tile_kernel<VPL, 1>), grid, block, 0, s, cost, left, out, disp, ws, A);
      else
        hipLaunchKernelGGL((sgm_tile_kernel<VPL, 2>), grid, block, 0, s, cost, left, out, disp, ws, A);
    }
    return check_launch("stm_sgm_f32 (tiled: tiles)");
  }
}

bool shape_ok(int H, int W, int D, int batch, int flags) {
  if (flags & (STM_SGM_SINGLE_PASSES | STM_SGM_EXACT_NONFINITE | STM_SGM_PAIRED)) return false;
  if (D != 64 && D != 128 && D != 256) return false;
  if (H < 1 || W < 1 || batch < 1 || batch > 65535) return false;
  if ((long long)batch * ((H + kT - 1) / kT) * ((W + kT - 1) / kT) > 0x3fffffffll) return false;
  return true;
}

}  // namespace

bool sgm_tiled_selected(int H, int W, int D, int batch, int flags) {
  if (!shape_ok(H, W, D, batch, flags)) return false;
  TiledPlan plan;
  if (!make_plan(H, W, D, batch, plan)) return false;
  if (!sweep_fits(D)) return false;
  if (flags & STM_SGM_TILED) return true;
  // Measured on MI355X (tools/sgm_forms.py): the tiled form wins where the volume traffic it
  // saves outweighs its fixed costs per image row and per tile, i.e. for wide disparity ranges
  // with enough columns in the batch to keep the row sweeps' column strips on most CUs:
  // D = 256: from 2 pairs at 1080p (10.8 against 11.3 ms for two, 4.5 against 5.7 ms per pair for
  // three, 4.0 against 5.8 at 8-16; 8.0 against 5.9 ms for a single pair); D = 128: from ~28 Teddy
  // pairs (5.6 against 5.3 ms at 24, 6.2 against 7.2 at 32, 42 against 55 at 256); D = 64: the paired
  // form is 1.4x faster and keeps the job.
  const int cus = device_cus();
  const int strip_cols = kSweepWaves * sweep_cw(D / 64);
  const long long need = (long long)cus * strip_cols * (D == 256 ? 9 : 15);  // tenths
  return D >= 128 && (long long)batch * W * 2 * 10 >= need && H >= 64;
}

bool sgm_tiled_aligned(const void *cost, const void *out, const void *workspace, int final_mode) {
  return (reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && (reinterpret_cast<uintptr_t>(cost) & 15) == 0 &&
         (reinterpret_cast<uintptr_t>(out) & 15) == 0;  // (`out` is read by STM_SGM_ACCUMULATE even when it is not written)
}

size_t sgm_tiled_workspace_bytes(int H, int W, int D, int batch, int flags) {
  if (!sgm_tiled_selected(H, W, D, batch, flags)) return 0;
  TiledPlan plan;
  make_plan(H, W, D, batch, plan);
  return plan.total_floats * sizeof(float);
}

int sgm_run_tiled(const float *cost, const float *left, float *out, int32_t *disp, int H, int W,
                  int D, float penalty1, float penalty2, int flags, void *workspace,
                  size_t workspace_bytes, int batch, hipStream_t stream, int final_mode) {
  TiledPlan plan;
  STM_REQUIRE(shape_ok(H, W, D, batch, flags) && make_plan(H, W, D, batch, plan), STM_EUNSUPPORTED,
              "stm_sgm_f32: the tiled form does not cover this call");
  const size_t need = plan.total_floats * sizeof(float);
  STM_REQUIRE(workspace && workspace_bytes >= need, STM_EWORKSPACE,
              "stm_sgm_f32: needs a workspace of %zu bytes (stm_sgm_workspace_bytes)", need);
  STM_REQUIRE(sgm_tiled_aligned(cost, out, workspace, final_mode), STM_EINVAL,
              "stm_sgm_f32: the tiled form needs 16-byte aligned volumes and workspace");
  const int dlim = (flags & STM_SGM_FULL_MIN) ? D : D - 1;
  const int true_paths = (flags & STM_SGM_TRUE_PATHS) ? 1 : 0;
  const int acc = (flags & STM_SGM_ACCUMULATE) ? 1 : 0;
  const bool coop = (flags & STM_SGM_COOPERATIVE) != 0 || getenv("STM_SWEEP_COOPERATIVE") != nullptr;
  float *ws = static_cast<float *>(workspace);
  switch (D / 64) {
    case 1: return run_tiled<1>(cost, left, out, disp, plan, penalty1, penalty2, dlim, true_paths, acc, coop, ws, batch, stream, final_mode);
    case 2: return run_tiled<2>(cost, left, out, disp, plan, penalty1, penalty2, dlim, true_paths, acc, coop, ws, batch, stream, final_mode);
    default: return run_tiled<4>(cost, left, out, disp, plan, penalty1, penalty2, dlim, true_paths, acc, coop, ws, batch, stream, final_mode);
  }
}

}  // namespace stm

#ifdef STM_TILE_STAMPS
extern "C" void stm_debug_set_tile_stamps(void *p) {
  stm::g_tile_stamps = static_cast<unsigned long long *>(p);
}
extern "C" void stm_debug_set_sweep_stamps(void *p) {
  stm::g_sweep_stamps = static_cast<unsigned long long *>(p);
}
#endif
