// Scanline dynamic programming: forward pass, end-disparity pick, backtrack.
//
// Reference semantics (CPU path): src/dynamic_programming.cu
//   forward   :24-68   per row: prev[d] initialised (transposed read, see below), then for
//                      col = 1..W-1, for d = 0..D-1 IN PLACE:
//                        c1 = d > 0 ? prev[d-1] (already updated) : inf; c2 = prev[d];
//                        c3 = d < D-1 ? prev[d+1] : inf
//                        (c1 < c2 && c1 < c3) -> (c1,-1); else c2 < c3 -> (c2, 0); else (c3,+1)
//                        path[row][col][d] = dir; prev[d] = cost[row][col][d] + m
//   end pick  :219     torch::argmin(row_final, 1): first minimum
//   backtrack :166-181 cur = end; disp[row][W-1] = cur; for col = W-2..0:
//                        cur = clamp(cur + path[row][col][cur], 0, D-1); disp[row][col] = cur
//
// Quirks kept in STM_DP_REFERENCE mode:
//   Q11a  the initial read is cost[d][row][0] (indices transposed)      -> needs D <= H <= W
//   Q11b  the initialisation writes path[d][row][0] = 0, which clobbers entries that
//         earlier rows already produced: after all rows, path[r][c][0] = 0 for
//         r < D, r < c < H.  Rows are independent otherwise, so the clobber is applied
//         as a separate tiny kernel after the forward pass instead of serialising rows.
//   Q12   the backtrack reads path[row][col] (not col+1) and so consumes column 0,
//         which the forward pass never computes; it is defined as zero here (the
//         reference leaves it to torch.empty; its pinned form zero-fills).
//
// The in-place d loop is a serial (min,+) chain with a rounded add at every link, so a
// scan would re-associate the adds.  Two exact parallelisations are used: the skewed
// wavefront of dp_tile_kernel (the fast path, below) and, for disparity counts it does not
// cover, a fixed-point iteration (dp_forward_kernel): with
// m23[d] = (c2 < c3 ? c2 : c3) the update is new[d] = cost[d] + (new[d-1] < m23[d] ?
// new[d-1] : m23[d]); starting from new[d] = cost[d] + m23[d] and re-applying that map
// until nothing changes reaches the chain's unique fixed point with the very same
// floating-point operations (each link is evaluated with its final inputs exactly
// once more than needed).  Chains are short in practice (a link is taken only where
// the lower neighbour's new cost undercuts both old costs), so a column costs a few
// wave-parallel sweeps instead of D dependent steps.
//
// dp_forward_kernel: one wave per image row; lane l holds disparities l*VPL .. l*VPL+VPL-1.
// Bound: HBM by bytes (4 read + 1 written per (pixel, d)); in practice the forward pass is
// latency-bound by its dependent chain.
#include <stdlib.h>

#include <atomic>

#include "stm_common.h"

namespace stm {
namespace {

constexpr int kDpBlock = 256;
constexpr int kDpPF = 4;

struct DpShape {
  int H, W, D, mode, batch;
};

template <int VPL, bool VEC>
__global__ __launch_bounds__(kDpBlock) void dp_forward_kernel(const float *__restrict__ vol_all,
                                                              int8_t *__restrict__ path_all,
                                                              float *__restrict__ rowfin_all,
                                                              const DpShape S) {
  const int lane = threadIdx.x & 63;
  const int wave = wave_uniform((int)(blockIdx.x * (kDpBlock / 64) + (threadIdx.x >> 6)));
  if (wave >= S.H * S.batch) return;
  const int b = wave / S.H;
  const int row = wave - b * S.H;
  const size_t npix = (size_t)S.H * S.W;
  const float *vol = vol_all + (size_t)b * npix * S.D;
  int8_t *path = path_all + (size_t)b * npix * S.D;
  float *rowfin = rowfin_all + ((size_t)b * S.H + row) * S.D;
  const int D = S.D, W = S.W;
  const int d0 = lane * VPL;
  const float inf = __builtin_inff();

  float prev[VPL];
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int d = d0 + j;
    float v = inf;
    if (d < D) {
      v = S.mode == STM_DP_REFERENCE ? vol[((size_t)d * W + row) * D]        // cost[d][row][0]
                                     : vol[((size_t)row * W) * D + d];      // cost[row][0][d]
      path[((size_t)row * W) * D + d] = 0;  // column 0 is defined as zero (Q12)
    }
    prev[j] = v;
  }

  // Ring of cost rows in flight.
  Row<VPL> cbuf[kDpPF];
  auto fetch = [&](int col, Row<VPL> &dst) {
    dst = load_row<VPL, VEC>(vol, (size_t)row * W + col, D, d0);
  };
#pragma unroll
  for (int u = 0; u < kDpPF; ++u) fetch(1 + u < W ? 1 + u : W - 1, cbuf[u]);

  for (int base = 1; base < W; base += kDpPF) {
#pragma unroll
    for (int u = 0; u < kDpPF; ++u) {
      const int col = base + u;
      if (col < W) {
        float c[VPL];
#pragma unroll
        for (int j = 0; j < VPL; ++j) c[j] = cbuf[u].v[j];
        fetch(col + kDpPF < W ? col + kDpPF : W - 1, cbuf[u]);

        // m23 and the +1 / 0 direction; prev[D] = +inf.
        const float nxt = lane_down(prev[0], inf);
        float m23[VPL];
        bool take2[VPL];
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
          const float c2 = prev[j];
          float c3 = j + 1 < VPL ? prev[j + 1] : nxt;
          if (d0 + j >= D - 1) c3 = inf;
          take2[j] = c2 < c3;
          m23[j] = take2[j] ? c2 : c3;
        }
        // Fixed-point iteration of the in-place chain.
        float nw[VPL];
#pragma unroll
        for (int j = 0; j < VPL; ++j) nw[j] = c[j] + m23[j];
        float left = inf;
        for (;;) {
          left = lane_up(nw[VPL - 1], inf);  // new[d0 - 1]; +inf for d0 == 0
          bool changed = false;
          float carry = left;
#pragma unroll
          for (int j = 0; j < VPL; ++j) {
            const float m = carry < m23[j] ? carry : m23[j];
            const float v = c[j] + m;
            // Bit compare: a NaN that reproduces itself must not keep the loop alive.
            changed |= __builtin_bit_cast(int, v) != __builtin_bit_cast(int, nw[j]);
            nw[j] = v;
            carry = v;
          }
          if (__ballot(changed) == 0ull) break;
        }
        // Directions from the final values.
        int8_t dir[VPL];
        {
          float carry = left;
#pragma unroll
          for (int j = 0; j < VPL; ++j) {
            dir[j] = carry < m23[j] ? (int8_t)-1 : (take2[j] ? (int8_t)0 : (int8_t)1);
            carry = nw[j];
          }
        }
        int8_t *pp = path + ((size_t)row * W + col) * D + d0;
        if (VPL == 4 && VEC) {
          if (d0 < D) {
            const unsigned w = (unsigned)(uint8_t)dir[0] | ((unsigned)(uint8_t)dir[1 % VPL] << 8) |
                               ((unsigned)(uint8_t)dir[2 % VPL] << 16) |
                               ((unsigned)(uint8_t)dir[3 % VPL] << 24);
            *reinterpret_cast<unsigned *>(pp) = w;
          }
        } else {
#pragma unroll
          for (int j = 0; j < VPL; ++j)
            if (d0 + j < D) pp[j] = dir[j];
        }
#pragma unroll
        for (int j = 0; j < VPL; ++j) prev[j] = (d0 + j < D) ? nw[j] : inf;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < VPL; ++j)
    if (d0 + j < D) rowfin[d0 + j] = prev[j];
}

// ---------------------------------------------------------------------------------
// Forward pass, skewed-wavefront form (the fast path; D a multiple of O, see the host code).
//
// Element (col, d) of a row depends on (col, d-1) [the in-place update], (col-1, d) and
// (col-1, d+1).  With the schedule t = O*col + d (O >= 2) all three are finished before step
// t, so the elements {(col, t - O*col)} are independent: a lane follows one column down the
// disparity axis -- its previous output stays in a register -- O steps behind the lane of
// the column to its left, whose outputs arrive through one DPP row rotation per step and
// wait in an O-deep register ring until they are (col-1, d) and (col-1, d+1).  This
// evaluates exactly the reference's serial recurrence (no re-association, no speculation)
// in O*(W-1) + D steps per row instead of W*D.
//
// Geometry: a DPP row of 16 lanes owns one image row, so a wave advances four image rows;
// P = 16*O >= D is the period a lane spends on a column, after which it continues with the
// column 16 to the right, i.e. the pipeline is filled once per row.  Time is cut in blocks
// of O steps: in block `blk` lane v works on the O-aligned chunk m = (blk - v) mod 16 of its
// column, so each lane needs exactly one aligned 4*O-byte piece of the cost volume per
// block.  The 64 pieces of the next block are fetched cooperatively as whole 4*O-byte runs
// (16-byte loads, 4*O/16 lanes per piece -- a lane-private stream would touch every cache
// line 4 to 8 times), parked in registers during the block and dropped into a padded LDS
// tile at its end; the path directions of a block leave as one O-byte store per lane.
// ---------------------------------------------------------------------------------
// Value of the lane to the left within the DPP row of 16 (row_ror:1).  Every lane has a
// source, so there is no "old" value to preserve: one v_mov_b32_dpp, no copy in front of it.
template <int R>
__device__ __forceinline__ float rot1(float x) {
  constexpr int ctrl = R == 16 ? 0x121 : 0x13C;  // row_ror:1 / wave_ror:1
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), ctrl, 0xf, 0xf, false));
}

// G: blocks whose cost pieces are fetched together, G * O * 4 = 128 bytes (one line) for
// O <= 16 when D is a multiple of 32 (G = 1: a piece per block, as above).  Pieces of 64 bytes
// and less, a kilobyte apart, are what the batch throughput hit at ~3 TB/s.
//
// R: lanes per image row.  16 (a DPP row, four image rows per wave) is the form described
// above.  64 (the whole wave on one image row, rotation by wave_ror:1) has R * O >= D with a
// quarter of the O, i.e. a quarter of the dependent steps per row and four times the waves: what
// a call with few rows wants (one 1080p pair is 270 waves of the 16-lane form, one per CU, each
// issuing its ~890 k instructions alone).  Pieces, parking and path runs are the same.
template <int O, int G, int R = 16, int NS = 2>
__global__ __launch_bounds__(64) void dp_tile_kernel(const float *__restrict__ vol_all,
                                                     int8_t *__restrict__ path_all,
                                                     float *__restrict__ rowfin_all,
                                                     const DpShape S) {
  static_assert(R == 16 || R == 64, "a DPP row or the whole wave");
  constexpr int RW = 64 / R;         // image rows per wave
  constexpr int LR = R == 16 ? 4 : 6;  // log2(R)
  constexpr int P = R * O;
  constexpr int LP = O / 4;          // load instructions per fetch; dwords of path bytes per block
  constexpr int LPG = G * O / 4;     // lanes per cost piece (16 bytes each)
  constexpr int PPI = 64 / LPG;      // pieces per cooperative load instruction
  constexpr int NGC = R / G;         // pieces per column (= piece-starting lanes per image row and block)
  constexpr int LD = G * O + 1;      // padded LDS row: conflict-free ds_read_b32 by 64 lanes
  // G = 1: two buffers (this block's pieces / the next block's).  G > 1: a lane's piece lasts G
  // blocks and its successor is parked at the end of the block that consumes its last part,
  // so one buffer does.
  __shared__ float tile[G == 1 ? 2 : 1][64 * LD];
  // Path bytes of the column each lane is working on, collected until a 64-byte run of the
  // column's record is complete (FL blocks of O bytes; the whole column for O >= 32), which
  // then leaves in one piece (see the end of the loop).  Keeping whole columns here (16 KB at
  // D = 256) made the LDS the limit on resident workgroups: four per CU; with runs it is
  // nine, which is what the kernel's throughput on a batch follows.
  constexpr int FL = O <= 16 ? 64 / O : 16;  // blocks per run
  static_assert(FL <= R && LP >= 1, "runs are whole blocks of one column");
  constexpr int PB = FL * LP + (O >= 32 ? 4 : 1);
  __shared__ __attribute__((aligned(16))) unsigned pbuf[64 * PB];
  // Column 0's initial costs (read in blocks 0-15) and the last column's final costs (written
  // from block W - 1 on) share their storage: the host takes this kernel only for W >= R + 1.  The
  // 4 KB this saves at D = 256 is what lets five workgroups instead of four share a CU's LDS
  // (30 of 160 KB each) -- the kernel's throughput at batch size is set by that number.
  __shared__ float initfin[RW][P + 1];
  float (&init)[RW][P + 1] = initfin;
  float (&fin)[RW][P + 1] = initfin;
  const int lane = threadIdx.x;
  const int rq = lane >> LR, v = lane & (R - 1);
  const int b = blockIdx.y;
  const int D = S.D, W = S.W, H = S.H;
  const int row = blockIdx.x * RW + rq;
  const bool rowok = row < H;
  const size_t npix = (size_t)H * W;
  const float *vol = vol_all + (size_t)b * npix * D;
  int8_t *path = path_all + (size_t)b * npix * D;
  const float inf = __builtin_inff();

  // Column 0: previous-cost initialisation (Q11a: the reference reads cost[d][row][0]).
  for (int d = v; d < D; d += R) {
    float x = 0.0f;
    if (rowok) x = S.mode == STM_DP_REFERENCE ? vol[((size_t)d * W + row) * D] : vol[((size_t)row * W) * D + d];
    init[rq][d] = x;
  }

  // Cooperative fetch of the cost pieces the next block will begin to consume.  At time T
  // (block T is the next one) the lanes v = T (mod G) of every image row start a new piece:
  // the G * O disparities from chunk T - v of their column walk (O disparities down the
  // column per block, 16 blocks per column, then the column 16 to the right).  Instruction u,
  // lane i serves the (u * PPI + i / LPG)-th of those 64 / G pieces, 16-byte part i % LPG.
  // Pieces without work read as zeros (out-of-range buffer loads); what their lanes compute
  // from them is read by lanes without work only (a lane's left neighbour starts earlier and
  // ends later than it does, and chunks past D - 1 are cut off by the c3 rule below).
  // Pieces and path runs go through buffer descriptors over this wave's image rows: a 32-bit
  // offset per lane, and an access without work is one with an out-of-range offset, which the
  // hardware drops -- no branch around a load or a store, so the compiler can count what is in
  // flight and a block waits for the pieces it parks only (with a branch it drained everything,
  // s_waitcnt vmcnt(0), the request of the same block included).
  typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
  typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
  constexpr unsigned kNowhere = 0x7fffffffu;
  const int row0 = blockIdx.x * RW;
  const unsigned rows_here = (unsigned)(H - row0 < RW ? H - row0 : RW);
  const BufRsrc crs = make_rsrc(vol + (size_t)row0 * W * D, rows_here * (unsigned)(W * D) * 4u);
  const BufRsrc prs = make_rsrc(path + (size_t)row0 * W * D, rows_here * (unsigned)(W * D));
  int T = 0;
  auto fetch = [&](float4 (&regs)[LP]) {
    const int set = T & (G - 1), n = T / G;
#pragma unroll
    for (int u = 0; u < LP; ++u) {
      const int j = u * PPI + lane / LPG, part = lane % LPG;
      const int pr = j / NGC, jj = j % NGC;  // image row of the wave, lane group
      const int pv = G * jj + set;
      const int k = n - jj;                       // the piece's index in its lane's walk
      const int col = pv + R * (k / NGC);         // (k >= 0 where it matters)
      const int pd0 = G * O * (k % NGC);
      const bool ok = (k >= 0) & (col < W) & (pd0 < D) & (row0 + pr < H);
      const unsigned at = ok ? 4u * (unsigned)((pr * W + col) * D + pd0 + 4 * part) : kNowhere;
      const v4u_t x = __builtin_amdgcn_raw_buffer_load_b128(crs, at, 0, 0);
      const unsigned x0 = x.x, x1 = x.y, x2 = x.z, x3 = x.w;  // (bit_cast of a vector element: see buf_load_row)
      regs[u] = make_float4(__builtin_bit_cast(float, x0), __builtin_bit_cast(float, x1),
                            __builtin_bit_cast(float, x2), __builtin_bit_cast(float, x3));
    }
    ++T;
  };
  // Parks what the fetch for time `when` brought.
  auto park = [&](int buf, const float4 (&regs)[LP], int when) {
    const int set = when & (G - 1);
#pragma unroll
    for (int u = 0; u < LP; ++u) {
      const int j = u * PPI + lane / LPG, part = lane % LPG;
      const int pi = (j / NGC) * R + G * (j % NGC) + set;   // the lane that owns the piece
      float *dst = &tile[buf][pi * LD + 4 * part];
      dst[0] = regs[u].x; dst[1] = regs[u].y; dst[2] = regs[u].z; dst[3] = regs[u].w;
    }
  };
  if (G > 1) {  // lanes whose first piece has not arrived yet read +inf
    for (int e = lane; e < 64 * LD; e += 64) tile[0][e] = inf;
    wave_lds_handoff();
  }

  // Two register sets: the pieces of block blk + 1 wait in one (requested during block blk - 1)
  // while those of block blk + 2 are requested into the other -- a block takes about a
  // microsecond, a load from HBM up to two or three, and with one block of lead every block of
  // a lone workgroup (a single pair: one per CU) waited for its loads.
  // (NS register sets, NS - 1 blocks of lead: the whole-wave form's blocks are a third as long.)
  static_assert(NS >= 2, "one set is parked while the others are in flight");
  float4 regs[NS][LP];
  fetch(regs[0]);
  park(0, regs[0], 0);
#pragma unroll
  for (int i = 1; i < NS; ++i) fetch(regs[i]);
  wave_lds_handoff();  // init[] and tile[0] are read by other lanes below
  float hist[O];
#pragma unroll
  for (int j = 0; j < O; ++j) hist[j] = inf;
  float prev = inf;
  const int nblocks = W + R - 1;
  auto block = [&](int blk, float4 (&freg)[LP], const float4 (&preg)[LP]) {
    const int cur = blk & 1;
    fetch(freg);  // the pieces of block blk + NS: in flight while this and the next NS - 1 blocks compute
    const int q = blk - v;
    const int col = v + R * (q >> LR);
    const int d0 = O * (q & (R - 1));
    const bool live = q >= 0 && col < W && d0 < D && rowok;
    // The block's O cost values move from LDS to registers up front, so the dependent
    // chain below has no memory waits.  Lanes without work get zero costs (see fetch): no
    // select in the chain, and nothing with work reads what they put out.
    float cst[O], outs[O];
#pragma unroll
    for (int j = 0; j < O; ++j) cst[j] = tile[G == 1 ? cur : 0][lane * LD + (q & (G - 1)) * O + j];
    unsigned pk[LP];
#pragma unroll
    for (int u = 0; u < LP; ++u) pk[u] = 0;
    const bool col0 = live && col == 0;
    if (blk < R) {
      // Column 0 is in flight (wave-uniform test): its "update" is the initial cost with
      // direction 0, so these blocks carry one extra select per step.
#pragma unroll
      for (int j = 0; j < O; ++j) {
        const int d = d0 + j;
        const float c2 = hist[j];                 // (col-1, d),   received O steps ago
        const float c3 = d < D - 1 ? hist[(j + 1) % O] : inf;  // (col-1, d+1)
        const float c1 = d > 0 ? prev : inf;
        const float m23 = c2 < c3 ? c2 : c3;
        const float m = c1 < m23 ? c1 : m23;
        const float out = col0 ? init[rq][d < P ? d : 0] : cst[j] + m;
        const int dir = col0 ? 0 : (c1 < m23 ? -1 : (c2 < c3 ? 0 : 1));
        prev = out;
        outs[j] = out;
        pk[j / 4] |= (unsigned)(dir & 0xff) << (8 * (j % 4));
        hist[j] = rot1<R>(out);        // row_ror:1 -> the column on the right
      }
    } else {
      // Steady state.  With m23 = (c2 < c3 ? c2 : c3) the reference's three-way choice is
      // new = cost + min(c1, m23), direction -1 iff c1 < m23: the dependent chain per step
      // is one v_min_f32 and one v_add_f32; m23 and the direction are off the chain.
      float c1 = d0 > 0 ? prev : inf;
#pragma unroll
      for (int j = 0; j < O; ++j) {
        const float c2 = hist[j];
        // Only the last element of a chunk can be d = D-1 (chunks are O-aligned, D % O == 0).
        const float c3 = (j < O - 1 || d0 + O < D) ? hist[(j + 1) % O] : inf;
        const float m23 = c2 < c3 ? c2 : c3;
        float t;  // v_min_f32 without the canonicalising v_max hipcc puts in front of fminf
        asm("v_min_f32 %0, %1, %2" : "=v"(t) : "v"(c1), "v"(m23));
        const float out = cst[j] + t;
        const int dir = c1 < m23 ? -1 : (c2 < c3 ? 0 : 1);
        outs[j] = out;
        pk[j / 4] |= (unsigned)(dir & 0xff) << (8 * (j % 4));
        hist[j] = rot1<R>(out);
        c1 = out;
      }
      prev = c1;
    }
    if (blk >= nblocks - R) {  // wave-uniform: the last column is in flight
      // Final costs of the row: the lane that owns the last column parks them in LDS.
      const bool isfin = live && col == W - 1;
#pragma unroll
      for (int j = 0; j < O; ++j) fin[rq][isfin ? d0 + j : P] = outs[j];  // slot P: dump
    }
    // Path directions.  A lane's O bytes of this block are one short piece of a column that
    // lies 4*D bytes away from its neighbours', so storing them directly costs 64 cache lines
    // per instruction (measured: more than half of the kernel at D = 128).  Instead they are
    // collected per column in LDS, FL blocks at a time (a 64-byte run of the column's record;
    // the whole column for O >= 32); in every block 16 / FL lanes per image row complete a run,
    // and those 4 x 16 / FL runs leave in one instruction, FL lanes x O bytes each.
    {
      unsigned *mine = &pbuf[lane * PB + (q & (FL - 1)) * LP];
#pragma unroll
      for (int u = 0; u < LP; ++u) mine[u] = pk[u];
      wave_lds_handoff();  // the finishing runs are read by other lanes
      // Lanes finishing a run in this block: those with (blk - lane) % FL == FL - 1, 16 / FL per
      // image row; lane v serves the (v / FL)-th of them, piece v % FL of its run.
      const int vd = ((blk + 1) & (FL - 1)) + FL * (v / FL);
      const int qd = blk - vd;                 // = FL - 1 (mod FL)
      const int cd = vd + R * (qd >> LR);      // that lane's column (negative before the first one)
      const int chunk = (qd & (R - 1)) - (FL - 1) + (v % FL);  // the piece's O disparities
      const unsigned *src = &pbuf[(rq * R + vd) * PB + (v % FL) * LP];
      unsigned out_w[LP];
#pragma unroll
      for (int u = 0; u < LP; ++u) out_w[u] = src[u];
      const bool okd = (qd >= 0) & (cd < W) & (row0 + rq < H) & (chunk * O < D);  // (this lane serves image row rq)
      const unsigned dst = okd ? (unsigned)((rq * W + cd) * D + chunk * O) : kNowhere;
#pragma unroll
      for (int u = 0; u < LP; u += 4) {
        if constexpr (LP >= 4) {
          v4u_t w4;
          w4.x = out_w[u]; w4.y = out_w[(u + 1) % LP]; w4.z = out_w[(u + 2) % LP]; w4.w = out_w[(u + 3) % LP];
          __builtin_amdgcn_raw_buffer_store_b128(w4, prs, dst, 4 * u, 0);
          asm volatile("s_nop 3" ::"v"(w4) : "memory");  // (hold_store_data, stm_common.h: the store still reads w4)
        } else if constexpr (LP == 2) {
          v2u_t w2;
          w2.x = out_w[0]; w2.y = out_w[1 % LP];
          __builtin_amdgcn_raw_buffer_store_b64(w2, prs, dst, 0, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b32(out_w[0], prs, dst, 0, 0);
        }
      }
    }
    park(G == 1 ? cur ^ 1 : 0, preg, blk + 1);
    wave_lds_handoff();  // the parked pieces (and pbuf) change hands between iterations
  };
  for (int blk = 0; blk < nblocks; blk += NS) {
    // (Whole groups of NS blocks, without a test between them: blocks past the last one find no
    // lane with work -- every column is >= W -- and the compiler can only count the loads and
    // stores in flight, i.e. wait for exactly the set a block parks, on straight-line code.)
#pragma unroll
    for (int i = 0; i < NS; ++i) block(blk + i, regs[i], regs[(i + 1) % NS]);
  }
  // Flush the final costs (fin[] was written by the lanes that own the last column).
  float *rowfin = rowfin_all + ((size_t)b * H + (rowok ? row : 0)) * D;
  if (rowok)
    for (int d = v; d < D; d += R) rowfin[d] = fin[rq][d];
}

// Q11b: path[r][c][0] = 0 for r < D, r < c < H (and c < W, implied by H <= W).
__global__ void dp_clobber_kernel(int8_t *__restrict__ path_all, const DpShape S) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  const int b = blockIdx.z;
  if (r < S.D && r < S.H && c > r && c < S.H && c < S.W)
    path_all[(((size_t)b * S.H + r) * S.W + c) * S.D] = 0;
}

// One wave per row: pick the end disparity, then walk back through the row's path
// records.  The records of a chunk of columns are pulled into LDS with coalesced
// loads; lane 0 follows the chain there and the chunk's disparities leave as one
// coalesced store.
constexpr int kBtChunk = 64;

__global__ __launch_bounds__(64) void dp_backtrack_kernel(const int8_t *__restrict__ path_all,
                                                          const float *__restrict__ rowfin_all,
                                                          int32_t *__restrict__ disp_all,
                                                          const DpShape S) {
  extern __shared__ __align__(16) unsigned char dp_smem[];
  int8_t *tile = reinterpret_cast<int8_t *>(dp_smem);                   // [kBtChunk][D]
  int *dchunk = reinterpret_cast<int *>(dp_smem + (size_t)kBtChunk * ((S.D + 15) & ~15));
  const int lane = threadIdx.x;
  const int row = blockIdx.x;
  const int b = blockIdx.y;
  const int D = S.D, W = S.W;
  const int8_t *path = path_all + (((size_t)b * S.H + row) * W) * D;
  const float *rf = rowfin_all + ((size_t)b * S.H + row) * D;
  int32_t *disp = disp_all + ((size_t)b * S.H + row) * W;

  // First minimum of the row's final costs (torch::argmin semantics for non-NaN input).
  float best = __builtin_inff();
  int idx = 0x7fffffff;
  for (int d = lane; d < D; d += 64) {
    const float q = rf[d];
    if (idx == 0x7fffffff) idx = d;
    if (q < best) { best = q; idx = d; }
  }
  const float m = group_min_f32<64>(best);
  int cand = (best == m) ? idx : 0x7fffffff;
  int cur = group_min_i32<64>(cand);
  if (cur == 0x7fffffff) cur = 0;  // all-NaN row
  if (lane == 0) disp[W - 1] = cur;

  // Columns W-2 .. 0 in chunks [lo, hi].
  for (int hi = W - 2; hi >= 0; hi -= kBtChunk) {
    const int lo = hi - kBtChunk + 1 > 0 ? hi - kBtChunk + 1 : 0;
    const int ncol = hi - lo + 1;
    const size_t bytes = (size_t)ncol * D;
    const int8_t *src = path + (size_t)lo * D;
    // Coalesced copy; 16-byte vectors when the chunk start is aligned.
    if ((((uintptr_t)src) & 15) == 0 && (bytes & 15) == 0) {
      const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
      uint4 *t4 = reinterpret_cast<uint4 *>(tile);
      for (size_t i = lane; i < bytes / 16; i += 64) t4[i] = s4[i];
    } else {
      for (size_t i = lane; i < bytes; i += 64) tile[i] = src[i];
    }
    __syncthreads();
    if (lane == 0) {
      for (int col = hi; col >= lo; --col) {
        cur = cur + (int)tile[(size_t)(col - lo) * D + cur];
        cur = cur < 0 ? 0 : (cur > D - 1 ? D - 1 : cur);
        dchunk[col - lo] = cur;
      }
    }
    __syncthreads();
    cur = __builtin_amdgcn_readfirstlane(cur);
    if (lane < ncol) disp[lo + lane] = dchunk[lane];
    __syncthreads();
  }
}

// The same walk reading only what it can reach: a path moves by at most one disparity per
// column, so over a chunk of K columns it stays within K of where it enters the chunk, and an
// aligned window of kBtWin = 2 K + 16 bytes per column holds every record it can touch.  A
// third of the path volume's bytes instead of all of them (the whole-chunk form reads N bytes
// to use H x W of them: 1.5 of 6.9 ms per eight 1080p pairs, at 2.8 TB/s), and a tenth of the
// LDS, so that every row of a batch is resident at once.  D a multiple of 16, D >= kBtWin,
// 16-byte-aligned records.
constexpr int kBtK = 32;
constexpr int kBtWin = 2 * kBtK + 16;  // 80

__global__ __launch_bounds__(64) void dp_backtrack_window_kernel(const int8_t *__restrict__ path_all,
                                                                 const float *__restrict__ rowfin_all,
                                                                 int32_t *__restrict__ disp_all,
                                                                 const DpShape S) {
  __shared__ __attribute__((aligned(16))) int8_t tile[kBtK * kBtWin];
  __shared__ int dchunk[kBtK];
  const int lane = threadIdx.x;
  const int row = blockIdx.x;
  const int b = blockIdx.y;
  const int D = S.D, W = S.W;
  const int8_t *path = path_all + (((size_t)b * S.H + row) * W) * D;
  const float *rf = rowfin_all + ((size_t)b * S.H + row) * D;
  int32_t *disp = disp_all + ((size_t)b * S.H + row) * W;

  // First minimum of the row's final costs (torch::argmin semantics for non-NaN input).
  float best = __builtin_inff();
  int idx = 0x7fffffff;
  for (int d = lane; d < D; d += 64) {
    const float q = rf[d];
    if (idx == 0x7fffffff) idx = d;
    if (q < best) { best = q; idx = d; }
  }
  const float m = group_min_f32<64>(best);
  int cand = (best == m) ? idx : 0x7fffffff;
  int cur = group_min_i32<64>(cand);
  if (cur == 0x7fffffff) cur = 0;  // all-NaN row
  if (lane == 0) disp[W - 1] = cur;

  constexpr int kParts = kBtWin / 16;  // 16-byte parts per column window
  for (int hi = W - 2; hi >= 0; hi -= kBtK) {
    const int lo = hi - kBtK + 1 > 0 ? hi - kBtK + 1 : 0;
    const int ncol = hi - lo + 1;
    int wlo = (cur - kBtK) & ~15;  // (cur - K may be negative: & ~15 rounds down)
    wlo = wlo < 0 ? 0 : (wlo > D - kBtWin ? D - kBtWin : wlo);
    for (int i = lane; i < ncol * kParts; i += 64) {
      const int c = i / kParts, part = i - c * kParts;
      reinterpret_cast<uint4 *>(tile)[i] =
          *reinterpret_cast<const uint4 *>(path + (size_t)(lo + c) * D + wlo + 16 * part);
    }
    __syncthreads();
    if (lane == 0) {
      for (int col = hi; col >= lo; --col) {
        cur = cur + (int)tile[(col - lo) * kBtWin + (cur - wlo)];
        cur = cur < 0 ? 0 : (cur > D - 1 ? D - 1 : cur);
        dchunk[col - lo] = cur;
      }
    }
    __syncthreads();
    cur = __builtin_amdgcn_readfirstlane(cur);
    if (lane < ncol) disp[lo + lane] = dchunk[lane];
    __syncthreads();
  }
}

// The windowed walk with the next chunk's records requested while this chunk is walked: the
// window is cut around the disparity at which the path enters the CURRENT chunk, so it must
// hold what the next chunk can reach from anywhere this one can leave it -- 2 K on either side,
// kBtWide = 4 K + 16 bytes per column.  1.8x the bytes of the plain windowed walk, which is why
// only calls with few rows take it: there a chunk is a load's latency plus a serial walk of K
// dependent LDS reads, and this form overlaps the two (one 1080p x 256 pair: 0.22 -> 0.13 ms).
constexpr int kBtWide = 4 * kBtK + 16;  // 144

__global__ __launch_bounds__(64) void dp_backtrack_ahead_kernel(const int8_t *__restrict__ path_all,
                                                                const float *__restrict__ rowfin_all,
                                                                int32_t *__restrict__ disp_all,
                                                                const DpShape S) {
  __shared__ __attribute__((aligned(16))) int8_t tile[2][kBtK * kBtWide];
  __shared__ int dchunk[kBtK];
  const int lane = threadIdx.x;
  const int row = blockIdx.x;
  const int b = blockIdx.y;
  const int D = S.D, W = S.W;
  const int8_t *path = path_all + (((size_t)b * S.H + row) * W) * D;
  const float *rf = rowfin_all + ((size_t)b * S.H + row) * D;
  int32_t *disp = disp_all + ((size_t)b * S.H + row) * W;

  // First minimum of the row's final costs (torch::argmin semantics for non-NaN input).
  float best = __builtin_inff();
  int idx = 0x7fffffff;
  for (int d = lane; d < D; d += 64) {
    const float q = rf[d];
    if (idx == 0x7fffffff) idx = d;
    if (q < best) { best = q; idx = d; }
  }
  const float m = group_min_f32<64>(best);
  int cand = (best == m) ? idx : 0x7fffffff;
  int cur = group_min_i32<64>(cand);
  if (cur == 0x7fffffff) cur = 0;  // all-NaN row
  if (lane == 0) disp[W - 1] = cur;

  constexpr int kParts = kBtWide / 16;                    // 16-byte parts per column window
  constexpr int kRegs = (kBtK * kParts + 63) / 64;        // parts per lane and chunk
  auto window = [&](int c) {                              // of a chunk entered at disparity c (and of its successor)
    int w = (c - 2 * kBtK) & ~15;                         // (negative: & ~15 rounds down)
    return w < 0 ? 0 : (w > D - kBtWide ? D - kBtWide : w);
  };
  uint4 regs[kRegs];
  auto request = [&](int hi, int wlo) {                   // the records of columns hi, hi - 1, .. (a chunk)
    const int lo = hi - kBtK + 1 > 0 ? hi - kBtK + 1 : 0;
    const int ncol = hi - lo + 1;
#pragma unroll
    for (int r = 0; r < kRegs; ++r) {
      const int i = lane + 64 * r;
      const int c = i / kParts, part = i - c * kParts;
      const bool ok = i < ncol * kParts;
      regs[r] = ok ? *reinterpret_cast<const uint4 *>(path + (size_t)(lo + c) * D + wlo + 16 * part) : make_uint4(0, 0, 0, 0);
    }
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int r = 0; r < kRegs; ++r) {
      const int i = lane + 64 * r;
      if (i < kBtK * kParts) reinterpret_cast<uint4 *>(tile[buf])[i] = regs[r];
    }
  };
  int wcur = window(cur);
  request(W - 2, wcur);
  park(0);
  __syncthreads();
  int buf = 0;
  for (int hi = W - 2; hi >= 0; hi -= kBtK) {
    const int lo = hi - kBtK + 1 > 0 ? hi - kBtK + 1 : 0;
    const int ncol = hi - lo + 1;
    const int wnext = window(cur);  // cut around this chunk's entry: holds the next chunk's reach
    if (lo > 0) request(lo - 1, wnext);
    if (lane == 0) {
      for (int col = hi; col >= lo; --col) {
        cur = cur + (int)tile[buf][(col - lo) * kBtWide + (cur - wcur)];
        cur = cur < 0 ? 0 : (cur > D - 1 ? D - 1 : cur);
        dchunk[col - lo] = cur;
      }
    }
    __syncthreads();
    cur = __builtin_amdgcn_readfirstlane(cur);
    if (lane < ncol) disp[lo + lane] = dchunk[lane];
    if (lo > 0) park(buf ^ 1);
    __syncthreads();
    buf ^= 1;
    wcur = wnext;
  }
}

// Whether a call takes the whole-wave form of the forward pass (tools/dp_scale.py): it has a
// quarter of the narrow form's dependent steps per wave on four times the waves and ~1.3x its
// instructions in total, so it wins while the narrow form leaves SIMDs without a wave -- one
// 1080p x 256 pair: 270 narrow waves on 1,024 SIMDs, 1.57 -> 1.00 ms; two 720p x 192 pairs
// 1.07 -> 0.68 ms; one or two Teddy pairs 0.26 -> 0.24 ms; from ~1.5 narrow waves per CU on the
// two take the same time and the narrow one is ahead on batches (16 x 1080p: 0.71 against 0.74 ms
// per pair; 16 Teddy pairs 0.033 against 0.045).  Above D = 256 the narrow form's O = 32 / 64
// blocks do not fit the register file (spills) and the whole-wave form is 2-3x faster at every
// batch size (1030 x 1040 x 1024: 6.2 -> 1.9 ms, four pairs 15.3 -> 6.0 ms).
bool dp_prefers_wide(int H, int D, int batch) {
  if (D > 256) return true;
  static std::atomic<int> cus_of[16];  // (any thread may call; a race stores the same value twice)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  int cus = dev >= 0 && dev < 16 ? cus_of[dev].load(std::memory_order_relaxed) : 0;
  if (cus == 0) {
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    if (dev >= 0 && dev < 16) cus_of[dev].store(cus, std::memory_order_relaxed);
  }
  const long long narrow_waves = (long long)batch * ((H + 3) / 4);
  return narrow_waves * 2 <= (long long)cus * 3;
}

}  // namespace
}  // namespace stm

extern "C" int stm_dp_f32(const float *volume, int8_t *path, float *row_final, int32_t *disparity,
                          int H, int W, int D, int mode, int batch, void *stream) {
  using namespace stm;
  STM_REQUIRE(volume && path && row_final && disparity, STM_EINVAL, "stm_dp_f32: null pointer");
  STM_REQUIRE(H > 0 && W > 0 && D > 0 && batch > 0, STM_EINVAL,
              "stm_dp_f32: non-positive size (H=%d W=%d D=%d batch=%d)", H, W, D, batch);
  STM_REQUIRE(mode == STM_DP_REFERENCE || mode == STM_DP_CORRECTED, STM_EINVAL,
              "stm_dp_f32: unknown mode %d", mode);
  STM_REQUIRE(D <= STM_MAX_DISPARITY, STM_EUNSUPPORTED, "stm_dp_f32: D=%d exceeds %d", D,
              STM_MAX_DISPARITY);
  if (mode == STM_DP_REFERENCE)
    STM_REQUIRE(D <= H && H <= W, STM_EINVAL,
                "stm_dp_f32: reference mode reads cost[d][row][0] and needs D <= H <= W "
                "(got H=%d W=%d D=%d); use STM_DP_CORRECTED", H, W, D);
  STM_REQUIRE(H <= 65535 && batch <= 65535, STM_EUNSUPPORTED, "stm_dp_f32: H or batch exceeds the grid limit");
  hipStream_t s = as_stream(stream);
  DpShape S;
  S.H = H; S.W = W; S.D = D; S.mode = mode; S.batch = batch;
  const int vpl = (D + 63) / 64;
  int VPL = 1;
  while (VPL < vpl) VPL <<= 1;
  const int waves = H * batch;
  const dim3 grid((waves + kDpBlock / 64 - 1) / (kDpBlock / 64)), block(kDpBlock);
  const uintptr_t align = VPL >= 4 ? 15 : (VPL == 2 ? 7 : 3);
  const bool vec = (D == VPL * 64) && ((reinterpret_cast<uintptr_t>(volume) & align) == 0) &&
                   ((reinterpret_cast<uintptr_t>(path) & 3) == 0);
  // Wavefront forms: whole O-disparity chunks (narrow: O = 8, 16, 32, 64 for D <= 128, 256, 512, 1024),
  // 16-byte-aligned cost rows and at least 17 columns (dp_tile_kernel's shared initial / final
  // cost buffer); everything else takes the fixed-point form.
  int O = 8;
  while (16 * O < D) O <<= 1;
  // (and four image rows of costs within the 2 GiB a 32-bit buffer offset is computed in)
  const bool wavefront = (D % O == 0) && W >= 17 && ((reinterpret_cast<uintptr_t>(volume) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(path) & 15) == 0) && (long long)W * D < (1ll << 27);
  // The whole-wave form (one image row per wave, O = 4 / 8 / 16 for D <= 256 / 512 / 1024): a
  // quarter of the dependent steps per row on four times the waves.
  int O64 = 4;
  while (64 * O64 < D) O64 <<= 1;
  const bool wide_ok = wavefront && D % 32 == 0 && D % O64 == 0 && W >= 65;
  bool wide = wide_ok && dp_prefers_wide(H, D, batch);
  if (const char *e = getenv("STM_DP_LANES")) wide = wide_ok && atoi(e) == 64;  // (A/B measurements)
  if (wide) {
    const dim3 wgrid(H, batch), wblock(64);
    if (O64 == 4) hipLaunchKernelGGL((dp_tile_kernel<4, 8, 64, 4>), wgrid, wblock, 0, s, volume, path, row_final, S);
    else if (O64 == 8) hipLaunchKernelGGL((dp_tile_kernel<8, 4, 64>), wgrid, wblock, 0, s, volume, path, row_final, S);
    else hipLaunchKernelGGL((dp_tile_kernel<16, 2, 64>), wgrid, wblock, 0, s, volume, path, row_final, S);
  } else if (wavefront) {
    const dim3 wgrid((H + 3) / 4, batch), wblock(64);
    const bool lines = D % 32 == 0;  // whole 128-byte pieces
    if (O == 8 && lines) hipLaunchKernelGGL((dp_tile_kernel<8, 4>), wgrid, wblock, 0, s, volume, path, row_final, S);
    else if (O == 8) hipLaunchKernelGGL((dp_tile_kernel<8, 1>), wgrid, wblock, 0, s, volume, path, row_final, S);
    else if (O == 16 && lines) hipLaunchKernelGGL((dp_tile_kernel<16, 2>), wgrid, wblock, 0, s, volume, path, row_final, S);
    else if (O == 16) hipLaunchKernelGGL((dp_tile_kernel<16, 1>), wgrid, wblock, 0, s, volume, path, row_final, S);
    else if (O == 32) hipLaunchKernelGGL((dp_tile_kernel<32, 1>), wgrid, wblock, 0, s, volume, path, row_final, S);
    else hipLaunchKernelGGL((dp_tile_kernel<64, 1>), wgrid, wblock, 0, s, volume, path, row_final, S);
  } else {
#define STM_DP_LAUNCH(V)                                                                              \
  do {                                                                                                \
    if (vec) hipLaunchKernelGGL((dp_forward_kernel<V, true>), grid, block, 0, s, volume, path, row_final, S); \
    else hipLaunchKernelGGL((dp_forward_kernel<V, false>), grid, block, 0, s, volume, path, row_final, S);    \
  } while (0)
  switch (VPL) {
    case 1: STM_DP_LAUNCH(1); break;
    case 2: STM_DP_LAUNCH(2); break;
    case 4: STM_DP_LAUNCH(4); break;
    case 8: STM_DP_LAUNCH(8); break;
    default: STM_DP_LAUNCH(16); break;
  }
#undef STM_DP_LAUNCH
  }
  int rc = check_launch("stm_dp_f32 (forward)");
  if (rc != STM_OK) return rc;
  if (mode == STM_DP_REFERENCE && H > 1) {
    const int rows = D < H ? D : H;
    hipLaunchKernelGGL(dp_clobber_kernel, dim3((H + 255) / 256, rows, batch), dim3(256), 0, s, path, S);
    rc = check_launch("stm_dp_f32 (clobber)");
    if (rc != STM_OK) return rc;
  }
  const size_t lds = (size_t)kBtChunk * ((D + 15) & ~15) + sizeof(int) * kBtChunk;
  // (The read-ahead form where the rows of the call leave most of the wave slots empty.)
  const bool few_rows = (long long)H * batch <= 4096;
  if (D % 16 == 0 && D >= kBtWide && few_rows && (reinterpret_cast<uintptr_t>(path) & 15) == 0)
    hipLaunchKernelGGL(dp_backtrack_ahead_kernel, dim3(H, batch), dim3(64), 0, s, path, row_final, disparity, S);
  else if (D % 16 == 0 && D >= kBtWin && (reinterpret_cast<uintptr_t>(path) & 15) == 0)
    hipLaunchKernelGGL(dp_backtrack_window_kernel, dim3(H, batch), dim3(64), 0, s, path, row_final, disparity, S);
  else
    hipLaunchKernelGGL(dp_backtrack_kernel, dim3(H, batch), dim3(64), lds, s, path, row_final, disparity, S);
  return check_launch("stm_dp_f32 (backtrack)");
}
