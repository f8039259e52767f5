// Body of the skewed-tile form's kernels, compiled once per disparity count (STM_SKEW_D = 128:
// stm_sgm_skew_d128.hip, 256: stm_sgm_skew_d256.hip); see stm_sgm_skew.hip for the form and
// stm_sgm_skew_index.h for the geometry.  No include guard: a translation unit includes it once.
#ifndef STM_SKEW_D
#error "define STM_SKEW_D (128 or 256) before including stm_sgm_skew_body.h"
#endif
#include "stm_sgm_tile_d128.h"
#include "stm_sgm_skew_index.h"
#include "stm_sgm_skew_launch.h"

namespace stm {
namespace {

using namespace skew;
using std::integral_constant;

#ifndef STM_SKEW_FWD_ENTRIES_AT
#define STM_SKEW_FWD_ENTRIES_AT 0
#endif
#ifndef STM_SKEW_FWD_PREFETCH_AT
#define STM_SKEW_FWD_PREFETCH_AT 2  // the next tile's cost rows are requested behind pair 1 / pair 2
#endif
#ifndef STM_SKEW_FWD_SPREAD
#define STM_SKEW_FWD_SPREAD 0
#endif
#ifndef STM_SKEW_COST_AUX
#define STM_SKEW_COST_AUX 2  // cache policy of the cost rows' loads / the side blocks' stores (2: non-temporal)
#endif
#ifndef STM_SKEW_SIDE_AUX
#define STM_SKEW_SIDE_AUX 2
#endif
#ifndef STM_SKEW_POLL_SLEEP
#define STM_SKEW_POLL_SLEEP 16  // s_sleep between two polls of a progress counter (units of 64 clocks)
#endif
#ifndef STM_SKEW_REV_OCC
#define STM_SKEW_REV_OCC 2  // workgroups of the reverse pass per CU (registers: 512 / (4 x this) per lane)
#endif
// Timing experiments (tools/build_skew_variants.sh; results are wrong with any of them):
#ifdef STM_SKEW_X_PLAIN
constexpr int kXSc1 = 0;
#else
constexpr int kXSc1 = kBufSc1;
#endif
#ifdef STM_SKEW_X_NOSTORE
#define STM_X_STORE(...) do {} while (0)
#else
#define STM_X_STORE(...) __VA_ARGS__
#endif
static_assert(kNowhere == kNoAccess, "one 'no access' offset");
constexpr int kD = STM_SKEW_D;              // 128: two chains per wave (half-waves), four waves per tile;
constexpr int kWaves = kD == 128 ? 4 : 8;    // 256: one chain per wave, eight waves per tile
constexpr int kThreads = kWaves * 64;
constexpr int kRowBytes = kD * 4;            // one pixel's disparities
constexpr int kSumPixBytes = st_stride(kD) * 4;
constexpr int kIpW = kT + 4;                 // intensity patch: u in [-2, 9], r in [-1, 8]
constexpr int kIpN = kIpW * (kT + 2);

struct SkewArgs {
  SkewGeom G;
  float p1, p2;
  int dlim;
  int true_paths;
#ifdef STM_SKEW_STAMPS
  unsigned long long *stamps;  // [tile column][wave][16] s_memtime stamps of one band of image 0
  int stamp_band;
#endif
};
// Diagnostic build (-DSTM_SKEW_STAMPS, tools/skew_stamps.py): where a tile's time goes.
#ifdef STM_SKEW_STAMPS
unsigned long long *g_skew_stamps = nullptr;
int g_skew_stamp_band = 0;
#define STM_SK_STAMP(k)                                                                                  \
  do {                                                                                                   \
    if (T.stamp && (threadIdx.x & 63) == 0)                                                              \
      P.stamps[((size_t)T.j * kWaves + (threadIdx.x >> 6)) * 16 + (k)] = __builtin_amdgcn_s_memtime();  \
  } while (0)
#else
#define STM_SK_STAMP(k) do {} while (0)
#endif

struct LaneId {
  int w, half, hl, d0, hl16;
  int row;  // the chain of the simple pairs (tile row of pair 0, tile column of pair 3), also the tile row this lane stages
  __device__ __forceinline__ void init() {
    const int lane = threadIdx.x & 63;
    w = wave_uniform((int)(threadIdx.x >> 6));
    if constexpr (kD == 128) {
      half = lane >> 5; hl = lane & 31; row = 2 * w + half;
    } else {
      half = 0; hl = lane; row = w;
    }
    d0 = hl * 4; hl16 = hl * 16;
  }
};

// The step engine: D = 128 two pixels per wave (sgm_step_h2*: the penalty is broadcast from a lane of
// `pen` inside the instruction), D = 256 one pixel per wave (sgm_step_fused; the penalty is read
// into a scalar).  LA / LB: the lanes of `pen` that hold the two chains' penalties of this step.
#if STM_SKEW_D == 128
typedef StepCtxH2 Ctx;
__device__ __forceinline__ void ctx_init(Ctx &X, float p1, const LaneId &L, int dlim) { X.init(p1, L.hl, dlim); }
template <int LA, int LB>
__device__ __forceinline__ void step2(Row<4> &pa, const Row<4> &ca, Row<4> &pb, const Row<4> &cb, float pen, Ctx &XA,
                                      Ctx &XB) {
  sgm_step_h2x2<LA, LB>(pa, ca, pb, cb, pen, XA, XB);
}
template <int LA>
__device__ __forceinline__ Row<4> step1(const Row<4> &p, const Row<4> &c, float pen, Ctx &X) {
  return sgm_step_h2<LA>(p, c, pen, X);
}
#else
typedef StepCtx Ctx;
__device__ __forceinline__ void ctx_init(Ctx &X, float p1, const LaneId &L, int dlim) {
  X.init(p1, L.d0, dlim, L.hl == 63 && dlim < kD);
}
template <int LA, int LB>
__device__ __forceinline__ void step2(Row<4> &pa, const Row<4> &ca, Row<4> &pb, const Row<4> &cb, float pen, Ctx &XA,
                                      Ctx &XB) {
  pa = sgm_step_fused<4>(pa, ca, read_lane(pen, LA), XA);
  pb = sgm_step_fused<4>(pb, cb, read_lane(pen, LB), XB);
}
template <int LA>
__device__ __forceinline__ Row<4> step1(const Row<4> &p, const Row<4> &c, float pen, Ctx &X) {
  return sgm_step_fused<4>(p, c, read_lane(pen, LA), X);
}
#endif

// Per-lane constants of a pair: the lane's chain.  Pieces are in SLOT order (which for pair 1's
// second half-wave is B before A, so that both halves of a wave change piece at the same slot).
// Per-lane constants of a pair (the lane's chain: skew::ChainIO, stm_sgm_skew_index.h).
typedef ChainIO PK;
template <int PHASE>
__device__ __forceinline__ void pk_init(PK &K, const LaneId &L) {
  chain_io_t<PHASE>(kD, L.w, L.half, L.d0, K);
}

// Forward pass: the chains' constants live in LDS, not in registers.  The kernel runs at the
// 256-register limit; with the ~20 constants in registers hipcc spills ~100 others around the tile
// loop, and a scratch access is a vector-memory access: its s_waitcnt vmcnt waits for every global
// load issued before it -- the entry states and the next tile's cost rows, requested a pair or two
// ahead ON PURPOSE -- so each spill reload on the way costs a memory round trip (s_memtime stamps,
// tools/skew_stamps.py: 15 of a tile's 43 thousand clocks).  An LDS read waits for LDS reads only.
// One row of kKtN words per (pair, half-wave), written once per workgroup from chain_io_t with
// d0 = 0; a lane's offset is the row's + 4 d0 (saturating: kNoAccess stays "no access").
constexpr int kKtN = 12;
constexpr int kKtWords = 4 * kWaves * 2 * kKtN;
template <int PHASE>
__device__ __forceinline__ void kt_fill(unsigned *Kt, const LaneId &L) {
  PK K;
  chain_io_t<PHASE>(kD, L.w, L.half, 0, K);
  if (L.hl == 0) {
    unsigned *row = Kt + ((PHASE * kWaves + L.w) * 2 + L.half) * kKtN;
    row[0] = (unsigned)K.pix[0]; row[1] = (unsigned)K.pix[1]; row[2] = (unsigned)K.pix[2];
    row[3] = (K.carry_first ? 1u : 0u) | (K.restart1 ? 2u : 0u);
    row[4] = K.ft; row[5] = K.fb; row[6] = 0u; row[7] = 0u;
    row[8] = K.ie[0]; row[9] = K.ie[1]; row[10] = K.ie[2]; row[11] = 0u;
  }
}
// A fetch of a pair's constants (what the caller does not use is not loaded); every fetch is a new
// one: nothing of an earlier fetch stays in registers for it.
template <int PHASE>
__device__ __forceinline__ PK kt_fetch(const unsigned *Kt, const LaneId &L) {
  unsigned o = (unsigned)(((PHASE * kWaves + L.w) * 2 + L.half) * kKtN);
  asm volatile("" : "+v"(o));
  const unsigned *row = Kt + o;
  const unsigned d = (unsigned)L.hl16;
  PK K;
  K.pix[0] = (int)row[0]; K.pix[1] = (int)row[1]; K.pix[2] = (int)row[2];
  const unsigned fl = row[3];
  K.carry_first = (fl & 1u) != 0u;
  K.restart1 = (fl & 2u) != 0u;
  K.e0_bottom = false;
  K.ft = __builtin_elementwise_add_sat(row[4], d);
  K.fb = __builtin_elementwise_add_sat(row[5], d);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    K.ie[q] = __builtin_elementwise_add_sat(row[8 + q], d);
    K.rs[q] = kNoAccess;
  }
  K.rb = K.rt = kNoAccess;
  return K;
}

// Where a tile's accesses go: descriptors over the image's cost volume, output and workspace
// block, and the tile's terms of the offsets (skew::TilePos).
struct TileCtx : TilePos {
  BufRsrc cost, out, wsb;
#ifdef STM_SKEW_STAMPS
  bool stamp;
#endif
};
// (the offsets themselves: row_state_off / side_state_off / fwd_inverse_entry_off, stm_sgm_skew_index.h)
__device__ __forceinline__ unsigned rowb_off(const SkewGeom &G, const TileCtx &T, int dir, int t, unsigned rel,
                                             bool need_row7) {
  return row_state_off(G, T, dir, t, rel, need_row7);
}
__device__ __forceinline__ unsigned side_off(const SkewGeom &G, const TileCtx &T, int jb, unsigned rel) {
  return side_state_off(G, T, jb, rel);
}
__device__ __forceinline__ unsigned fwd_ie_off(const SkewGeom &G, const TileCtx &T, int dir, unsigned e) {
  return fwd_inverse_entry_off(G, T, dir, e);
}

// ---- cost tile, intensities ---------------------------------------------------------------------
// Half-wave (w, half) stages tile row r = 2 w + half: pixels u = 0..7 at x = x0 + u - r, y = y0 + r,
// clamped into the image (a clamped pixel is never used).
struct Pre {
  Row<4> cpre[kT];
  float ipre;
};
__device__ __forceinline__ void prefetch_cost(Pre &P, const SkewGeom &G, const TileCtx &T, const float *image,
                                              const LaneId &L) {
  const int r = L.row;
  if (!T.gen) {
    const unsigned base = ((unsigned)((T.y0 + r) * G.W + T.x0 - r) * (unsigned)kD + (unsigned)L.d0) * 4u;
#pragma unroll
    for (int u = 0; u < kT; ++u) P.cpre[u] = buf_load_row4<STM_SKEW_COST_AUX>(T.cost, base + (unsigned)(u * kRowBytes), 0);
  } else {
    const int y = imin_(T.y0 + r, G.H - 1);
#pragma unroll
    for (int u = 0; u < kT; ++u) {
      const int x = imin_(imax_(T.x0 + u - r, 0), G.W - 1);
      P.cpre[u] = buf_load_row4<STM_SKEW_COST_AUX>(T.cost, ((unsigned)(y * G.W + x) * (unsigned)kD + (unsigned)L.d0) * 4u, 0);
    }
  }
  const int e = imin_((int)threadIdx.x, kIpN - 1);
  const int rr = e / kIpW - 1, uu = e % kIpW - 2;
  const int y = imin_(imax_(T.y0 + rr, 0), G.H - 1), x = imin_(imax_(T.x0 + uu - rr, 0), G.W - 1);
  P.ipre = image[y * G.W + x];
}
// ... one row of it at a time (the forward pass spreads the next tile's requests over the steps of a
// pair: thirteen vector-memory instructions issued back to back by all four waves at once wait
// for the CU's one address unit, and a wave that cannot issue does nothing else either).
template <int U>
__device__ __forceinline__ void prefetch_cost_row(Pre &P, const SkewGeom &G, const TileCtx &T, const LaneId &L) {
  const int r = L.row;
  unsigned off;
  if (!T.gen) {
    off = ((unsigned)((T.y0 + r) * G.W + T.x0 - r) * (unsigned)kD + (unsigned)L.d0) * 4u + (unsigned)(U * kRowBytes);
  } else {
    const int y = imin_(T.y0 + r, G.H - 1);
    const int x = imin_(imax_(T.x0 + U - r, 0), G.W - 1);
    off = ((unsigned)(y * G.W + x) * (unsigned)kD + (unsigned)L.d0) * 4u;
  }
  P.cpre[U] = buf_load_row4<STM_SKEW_COST_AUX>(T.cost, off, 0);
}
__device__ __forceinline__ void prefetch_patch(Pre &P, const SkewGeom &G, const TileCtx &T, const float *image) {
  const int e = imin_((int)threadIdx.x, kIpN - 1);
  const int rr = e / kIpW - 1, uu = e % kIpW - 2;
  const int y = imin_(imax_(T.y0 + rr, 0), G.H - 1), x = imin_(imax_(T.x0 + uu - rr, 0), G.W - 1);
  P.ipre = image[y * G.W + x];
}
__device__ __forceinline__ void stage_cost(const Pre &P, float *Ct, float *Ip, const LaneId &L) {
  const int r = L.row;
#pragma unroll
  for (int u = 0; u < kT; ++u) h2_lds_store(Ct, (r * kT + u) * kRowBytes + L.hl16, P.cpre[u]);
  if (threadIdx.x < kIpN) Ip[threadIdx.x] = P.ipre;
}

// Penalties of a chain: lane l of every 16-lane row holds the one of slot l & 7 -- lanes 0-7 of
// chain A (phase PA, direction sign SA: -1 = the step from the forward predecessor, +1 = from the
// forward successor, i.e. the inverse chain), lanes 8-15 of chain B.
template <int PA, int SA, int PB, int SB>
__device__ __forceinline__ float chain_penalties(const PK &KA, int naA, int nbA, const PK &KB, int naB, int nbB,
                                                 const float *Ip, float p1, float p2, int lane) {
  const int s = lane & 7;
  const bool second = (lane >> 3) & 1;
  const int na = second ? naB : naA, nb = second ? nbB : nbA;
  const int q = s < na ? 0 : (s < nb ? 1 : 2);
  const int start = q == 0 ? 0 : (q == 1 ? na : nb);
  const int pixA = KA.pix[0] * (q == 0) + KA.pix[1] * (q == 1) + KA.pix[2] * (q == 2) + (s - start) * tile_pstep(PA);
  const int pixB = KB.pix[0] * (q == 0) + KB.pix[1] * (q == 1) + KB.pix[2] * (q == 2) + (s - start) * tile_pstep(PB);
  const int pix = second ? pixB : pixA;
  const int du = second ? SB * tile_du(PB) : SA * tile_du(PA), dr = second ? SB * tile_dr(PB) : SA * tile_dr(PA);
  const int u = pix & (kT - 1), r = pix >> 3;
  return sgm_penalty(Ip[(r + 1) * kIpW + u + 2], Ip[(r + dr + 1) * kIpW + u + du + 2], p1, p2);
}
// The same in two halves: where the lane's two samples lie in the intensity patch (byte offsets,
// 16 bits each) is the same for every tile -- computed once per kernel; per tile two LDS reads and
// the division are left (a third of the instructions).
template <int PA, int SA, int PB, int SB>
__device__ __forceinline__ unsigned chain_penalty_offsets(const PK &KA, int naA, int nbA, const PK &KB, int naB,
                                                          int nbB, int lane) {
  const int s = lane & 7;
  const bool second = (lane >> 3) & 1;
  const int na = second ? naB : naA, nb = second ? nbB : nbA;
  const int q = s < na ? 0 : (s < nb ? 1 : 2);
  const int start = q == 0 ? 0 : (q == 1 ? na : nb);
  const int pixA = KA.pix[0] * (q == 0) + KA.pix[1] * (q == 1) + KA.pix[2] * (q == 2) + (s - start) * tile_pstep(PA);
  const int pixB = KB.pix[0] * (q == 0) + KB.pix[1] * (q == 1) + KB.pix[2] * (q == 2) + (s - start) * tile_pstep(PB);
  const int pix = second ? pixB : pixA;
  const int du = second ? SB * tile_du(PB) : SA * tile_du(PA), dr = second ? SB * tile_dr(PB) : SA * tile_dr(PA);
  const int u = pix & (kT - 1), r = pix >> 3;
  return (unsigned)(((r + 1) * kIpW + u + 2) * 4) | ((unsigned)(((r + dr + 1) * kIpW + u + du + 2) * 4) << 16);
}
__device__ __forceinline__ float penalty_at(const float *Ip, unsigned o, float p1, float p2) {
  const char *b = reinterpret_cast<const char *>(Ip);
  return sgm_penalty(*reinterpret_cast<const float *>(b + (o & 0xffffu)), *reinterpret_cast<const float *>(b + (o >> 16)), p1, p2);
}

// Gen tiles: where the lane's chain BEGINS a path (L = C): bit s of `bf` / `bi`: slot s lies in the
// image and its forward predecessor / successor does not.  Inside a piece the neighbour is the
// neighbouring slot; at a piece's ends it lies in another tile.
__device__ __forceinline__ bool in_image_u(const SkewGeom &G, int x, int y) {
  return (unsigned)x < (unsigned)G.W && (unsigned)y < (unsigned)G.H;
}
template <int PHASE, int NA, int NB>
__device__ __forceinline__ void begin_masks(const PK &K, const SkewGeom &G, const TileCtx &T, unsigned &bf, unsigned &bi) {
  constexpr int NP = NB < kT ? 3 : (NA < kT ? 2 : 1);
  constexpr int dx = img_dx(PHASE), dy = img_dy(PHASE);
  unsigned v = 0, pv = 0, sv = 0;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int st = q == 0 ? 0 : (q == 1 ? NA : NB), en = q == 0 ? NA : (q == 1 ? NB : kT);
    const int u = K.pix[q] & (kT - 1), r = K.pix[q] >> 3;
    const int x = T.x0 + u - r, y = T.y0 + r;  // the piece's first pixel
#pragma unroll
    for (int t = 0; t < kT; ++t) {
      if (st + t < en) v |= in_image_u(G, x + t * dx, y + t * dy) ? (1u << (st + t)) : 0u;
    }
    pv |= in_image_u(G, x - dx, y - dy) ? (1u << st) : 0u;
    sv |= in_image_u(G, x + (en - st) * dx, y + (en - st) * dy) ? (1u << (en - 1)) : 0u;
  }
  constexpr unsigned starts = 1u | (NA < kT ? 1u << NA : 0u) | (NB < kT ? 1u << NB : 0u);
  constexpr unsigned ends = (1u << (NA - 1)) | (NA < kT ? 1u << (NB - 1) : 0u) | (1u << (kT - 1));
  pv |= (v << 1) & ~starts;
  sv |= (v >> 1) & ~ends;
  bf = v & ~pv;
  bi = v & ~sv;
}
__device__ __forceinline__ Row<4> sel_row(bool c, const Row<4> &a, const Row<4> &b) {
  Row<4> r;
#pragma unroll
  for (int k = 0; k < 4; ++k) r.v[k] = c ? a.v[k] : b.v[k];
  return r;
}

// =====================================================================================
// Forward pass: one pair of a tile for one wave (two chains): both chains of the pair stepped
// together (slot s of the forward chain with slot 7 - s of the inverse chain), the update
// S = ((S + [I x pre]) + F) + I x inv of the sum tile from the fifth step on (stm_sgm_tile_h2.hip).
//   Ef / Ei   entry states per piece (slot order); Xf: the forward chain's state after each piece.
//   cls       count class of the tile for this pair (skew::count_class).
// =====================================================================================
template <int PHASE, int NA, int NB, int FINAL, bool GEN, typename Hook>
__device__ __forceinline__ void fwd_pair(const PK &K, const Row<4> (&Ef)[3], const Row<4> (&Ei)[3], Row<4> (&Xf)[3],
                                         int cls, bool sp, const SkewGeom &G, const TileCtx &T, int true_paths,
                                         const float *Ct, float *St, float pen, Ctx &X, const LaneId &L,
                                         Hook hook) {
  // hook(s): called in front of step s; the one of step kT / 2 holds the workgroup's barrier behind
  // which the sum tile may be read and updated.
  constexpr int pstep = tile_pstep(PHASE);
  constexpr int NP = NB < kT ? 3 : (NA < kT ? 2 : 1);
  Ctx XI = X;
  const int hl16 = L.hl16;
  const bool run_inv = cls != 0;
  auto pix_of = [&](auto s_) {
    constexpr int s = decltype(s_)::value;
    constexpr int q = s < NA ? 0 : (s < NB ? 1 : 2), st = q == 0 ? 0 : (q == 1 ? NA : NB);
    return K.pix[q] + (s - st) * pstep;
  };
  auto addr = [&](auto s_) { return pix_of(s_) * kRowBytes + hl16; };
  auto saddr = [&](auto s_) { return pix_of(s_) * kSumPixBytes + hl16; };
  // Whether the inverse pass of the lane's pieces counts at all (cls >= 2), from each piece's first
  // pixel.  A line that counts several times, or before its forward pass, is a DEFERRED line
  // (stm_sgm_skew_index.h): its pixels get the plain update here -- a value nobody uses -- and
  // their real sum from finish_deferred.
  int inv[3] = {1, 1, 1};
  if (PHASE >= 2 && cls >= 2) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int u = K.pix[q] & (kT - 1), r = K.pix[q] >> 3;
      inv[q] = line_counts(PHASE, T.x0 + u - r, T.y0 + r, G.H, G.W, true_paths).inv;
    }
  }
  Row<4> LF[kT], LI[kT];
  Row<4> pf = Ef[0];
  Row<4> pi = Ei[NP - 1];
  unsigned bf = 0, bi = 0;
  if constexpr (GEN) begin_masks<PHASE, NA, NB>(K, G, T, bf, bi);
  auto update = [&](auto s_, Row<4> &acc) {
    constexpr int s = decltype(s_)::value;
    constexpr int q = s < NA ? 0 : (s < NB ? 1 : 2);
    if (PHASE == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc.v[k] = 0.0f;
    }
    if (PHASE >= 2 && sp) {
      // A tile with deferred pixels: what their sum is made of goes to the band's block (rows 0-2:
      // the sum before pair 2 and pair 2's values; rows 3, 4: pair 3's); every other lane's stores
      // have no address.
      const int pix = pix_of(s_);
      const int x = T.x0 + (pix & (kT - 1)) - (pix >> 3), y = T.y0 + (pix >> 3);
      const int e = in_image(G, x, y) ? defer_entry(G, true_paths, x, y) : -1;
      const unsigned o = e >= 0 ? (unsigned)(defer_in_img(G, T.i, e, PHASE == 2 ? 0 : 3) * 4) + (unsigned)L.d0 * 4u : kNowhere;
      if (PHASE == 2) {
        buf_store_row4<kBufSc1>(T.wsb, o, 0, acc);
        buf_store_row4<kBufSc1>(T.wsb, o, kRowBytes, LF[s]);
        buf_store_row4<kBufSc1>(T.wsb, o, 2 * kRowBytes, LI[s]);
      } else {
        buf_store_row4<kBufSc1>(T.wsb, o, 0, LF[s]);
        buf_store_row4<kBufSc1>(T.wsb, o, kRowBytes, LI[s]);
      }
    }
    f32x2 lo = f32x2{acc.v[0], acc.v[1]}, hi = f32x2{acc.v[2], acc.v[3]};
    const f32x2 ilo = f32x2{LI[s].v[0], LI[s].v[1]}, ihi = f32x2{LI[s].v[2], LI[s].v[3]};
    lo = lo + f32x2{LF[s].v[0], LF[s].v[1]};
    hi = hi + f32x2{LF[s].v[2], LF[s].v[3]};
    if (run_inv) {
      const f32x2 lo2 = lo + ilo, hi2 = hi + ihi;
      const bool r = PHASE < 2 || inv[q] != 0;
      lo = r ? lo2 : lo;
      hi = r ? hi2 : hi;
    }
    acc.v[0] = lo.x; acc.v[1] = lo.y; acc.v[2] = hi.x; acc.v[3] = hi.y;
    if (PHASE < 3 || FINAL != 0) h2_lds_store(St, saddr(s_), acc);
  };
  // Pair 3: the finished sum of a slot leaves (unless only the winners are wanted).
  auto finish = [&](auto s_, const Row<4> &acc) {
    if constexpr (PHASE == 3 && FINAL != 2) {
      const int pix = pix_of(s_);
      const int u = pix & (kT - 1), r = pix >> 3;
      const int x = T.x0 + u - r, y = T.y0 + r;
      const bool keep = (!GEN || in_image(G, x, y)) && !(sp && defer_entry(G, true_paths, x, y) >= 0);
      const unsigned off = keep ? ((unsigned)(y * G.W + x) * (unsigned)kD + (unsigned)L.d0) * 4u : kNowhere;
      buf_store_row4<0>(T.out, off, 0, acc);
    }
  };
  auto sum_of = [&](auto s_) {
    Row<4> r;
    if (PHASE != 0) r = h2_lds_row(St, saddr(s_));
    else r = Ef[0];  // (unused: any defined value)
    return r;
  };
  auto note_exit = [&](auto s_) {
    constexpr int s = decltype(s_)::value;
    if constexpr (s == NA - 1) Xf[0] = pf;
    if constexpr (NA < kT ? s == NB - 1 : s == kT - 1) Xf[1] = pf;  // (a piece the chain does not have: a defined value)
    if constexpr (s == kT - 1) Xf[2] = pf;
  };
  if (run_inv) {
    Row<4> cf = h2_lds_row(Ct, addr(integral_constant<int, 0>()));
    Row<4> ci = h2_lds_row(Ct, addr(integral_constant<int, kT - 1>()));
    Row<4> sa, sb;
    auto both = [&](auto s_) {
      constexpr int s = decltype(s_)::value, r = kT - 1 - s;
      if constexpr (NA < kT && s == NA) pf = PHASE == 1 ? sel_row(K.restart1, Ef[1], pf) : Ef[1];
      if constexpr (NB < kT && s == NB) pf = Ef[2];
      if constexpr (NB < kT && r == NB - 1) pi = Ei[1];
      if constexpr (NA < kT && r == NA - 1) pi = PHASE == 1 ? sel_row(K.restart1, Ei[0], pi) : Ei[0];
      Row<4> cfn = cf, cin = ci;
      if constexpr (s < kT - 1) {
        cfn = h2_lds_row(Ct, addr(integral_constant<int, s + 1>()));
        cin = h2_lds_row(Ct, addr(integral_constant<int, r - 1>()));
      }
      hook(s_);
      if constexpr (s == kT / 2) {
        sa = sum_of(integral_constant<int, s>());
        sb = sum_of(integral_constant<int, r>());
      }
      step2<s, 8 + r>(pf, cf, pi, ci, pen, X, XI);
      if constexpr (GEN) {
        pf = sel_row(((bf >> s) & 1u) != 0u, cf, pf);
        pi = sel_row(((bi >> r) & 1u) != 0u, ci, pi);
      }
      LF[s] = pf;
      LI[r] = pi;
      note_exit(s_);
      cf = cfn;
      ci = cin;
      if constexpr (s >= kT / 2) {
        update(integral_constant<int, s>(), sa);
        finish(integral_constant<int, s>(), sa);
        update(integral_constant<int, r>(), sb);
        finish(integral_constant<int, r>(), sb);
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
  } else {
    // No inverse pass counts anywhere in this tile (Q6): the forward chain alone.
    hook(integral_constant<int, kT / 2>());
    Row<4> sa = sum_of(integral_constant<int, 0>());
    auto fstep = [&](auto s_) {
      constexpr int s = decltype(s_)::value;
      if constexpr (s != kT / 2) hook(s_);
      if constexpr (NA < kT && s == NA) pf = PHASE == 1 ? sel_row(K.restart1, Ef[1], pf) : Ef[1];
      if constexpr (NB < kT && s == NB) pf = Ef[2];
      const Row<4> c = h2_lds_row(Ct, addr(s_));
      pf = step1<s>(pf, c, pen, X);
      if constexpr (GEN) pf = sel_row(((bf >> s) & 1u) != 0u, c, pf);
      LF[s] = pf;
      LI[s] = pf;  // (not added: inv == 0)
      note_exit(s_);
      update(s_, sa);
      finish(s_, sa);
      if constexpr (s < kT - 1) sa = sum_of(integral_constant<int, s + 1>());
    };
    fstep(integral_constant<int, 0>()); fstep(integral_constant<int, 1>());
    fstep(integral_constant<int, 2>()); fstep(integral_constant<int, 3>());
    fstep(integral_constant<int, 4>()); fstep(integral_constant<int, 5>());
    fstep(integral_constant<int, 6>()); fstep(integral_constant<int, 7>());
  }
}

template <int PHASE, int FINAL, bool GEN, typename Hook>
__device__ __forceinline__ void fwd_pair_w(int w, const PK &K, const Row<4> (&Ef)[3], const Row<4> (&Ei)[3],
                                           Row<4> (&Xf)[3], int cls, bool sp, const SkewGeom &G, const TileCtx &T,
                                           int true_paths, const float *Ct, float *St, float pen, Ctx &X,
                                           const LaneId &L, Hook hook) {
#define STM_FWD_PAIR(NA, NB) \
  fwd_pair<PHASE, NA, NB, FINAL, GEN>(K, Ef, Ei, Xf, cls, sp, G, T, true_paths, Ct, St, pen, X, L, hook)
  if constexpr (PHASE == 0 || PHASE == 3) {
    STM_FWD_PAIR(kT, kT);
  } else if constexpr (PHASE == 1 && kD == 128) {
    switch (w) {
      case 0: STM_FWD_PAIR(7, kT); break;
      case 1: STM_FWD_PAIR(6, kT); break;
      case 2: STM_FWD_PAIR(5, kT); break;
      default: STM_FWD_PAIR(4, kT); break;
    }
  } else if constexpr (PHASE == 1) {
    switch (w) {
      case 0: STM_FWD_PAIR(kT, kT); break;
      case 1: STM_FWD_PAIR(7, kT); break;
      case 2: STM_FWD_PAIR(6, kT); break;
      case 3: STM_FWD_PAIR(5, kT); break;
      case 4: STM_FWD_PAIR(4, kT); break;
      case 5: STM_FWD_PAIR(3, kT); break;
      case 6: STM_FWD_PAIR(2, kT); break;
      default: STM_FWD_PAIR(1, kT); break;
    }
  } else {
    switch (kD == 128 ? w : w >> 1) {  // (D = 256: waves 2 k and 2 k + 1 own chains 7 - 2 k and 6 - 2 k, same pieces)
      case 0: STM_FWD_PAIR(1, 5); break;
      case 1: STM_FWD_PAIR(2, 6); break;
      case 2: STM_FWD_PAIR(3, 7); break;
      default: STM_FWD_PAIR(4, kT); break;
    }
  }
#undef STM_FWD_PAIR
}
// Slots at which wave w's chains of a pair change piece.
__device__ __forceinline__ int na_of(int phase, int w) { return skew::na_of(kD, phase, w); }
__device__ __forceinline__ int nb_of(int phase, int w) { return skew::nb_of(kD, phase, w); }

// =====================================================================================
// Reverse pass: the inverse chains of TWO pairs of a tile for one wave, stepped together (slot
// 7 - s of both).  Ea / Eb: entry states per piece; Xa / Xb: the chain's state after each piece
// (in slot order: X[q] is the state after the piece's FIRST slot, where the inverse chain leaves it).
// =====================================================================================
template <int PA, int NAA, int NBA, int PB, int NAB, int NBB>
__device__ __forceinline__ void rev_pairs(const PK &KA, const Row<4> (&Ea)[3], Row<4> (&Xa)[3], const PK &KB,
                                          const Row<4> (&Eb)[3], Row<4> (&Xb)[3], const SkewGeom &G,
                                          const TileCtx &T, const float *Ct, float pen, Ctx &X,
                                          const LaneId &L) {
  constexpr int NPA = NBA < kT ? 3 : (NAA < kT ? 2 : 1), NPB = NBB < kT ? 3 : (NAB < kT ? 2 : 1);
  Ctx XB = X;
  const int hl16 = L.hl16;
  auto pix_a = [&](auto s_) {
    constexpr int s = decltype(s_)::value;
    constexpr int q = s < NAA ? 0 : (s < NBA ? 1 : 2), st = q == 0 ? 0 : (q == 1 ? NAA : NBA);
    return KA.pix[q] + (s - st) * tile_pstep(PA);
  };
  auto pix_b = [&](auto s_) {
    constexpr int s = decltype(s_)::value;
    constexpr int q = s < NAB ? 0 : (s < NBB ? 1 : 2), st = q == 0 ? 0 : (q == 1 ? NAB : NBB);
    return KB.pix[q] + (s - st) * tile_pstep(PB);
  };
  unsigned bia = 0, bib = 0, unused_f;
  // (one copy of the code for clipped and full tiles: in the reverse pass two instantiations side
  // by side bought nothing)
  const bool gen = T.gen;
  if (gen) {
    begin_masks<PA, NAA, NBA>(KA, G, T, unused_f, bia);
    begin_masks<PB, NAB, NBB>(KB, G, T, unused_f, bib);
  }
  Row<4> pa = Ea[NPA - 1], pb = Eb[NPB - 1];
  Row<4> ca = h2_lds_row(Ct, pix_a(integral_constant<int, kT - 1>()) * kRowBytes + hl16);
  Row<4> cb = h2_lds_row(Ct, pix_b(integral_constant<int, kT - 1>()) * kRowBytes + hl16);
  auto both = [&](auto r_) {
    constexpr int r = decltype(r_)::value;
    if constexpr (NBA < kT && r == NBA - 1) pa = Ea[1];
    if constexpr (NAA < kT && r == NAA - 1) pa = PA == 1 ? sel_row(KA.restart1, Ea[0], pa) : Ea[0];
    if constexpr (NBB < kT && r == NBB - 1) pb = Eb[1];
    if constexpr (NAB < kT && r == NAB - 1) pb = PB == 1 ? sel_row(KB.restart1, Eb[0], pb) : Eb[0];
    Row<4> can = ca, cbn = cb;
    if constexpr (r > 0) {
      can = h2_lds_row(Ct, pix_a(integral_constant<int, r - 1>()) * kRowBytes + hl16);
      cbn = h2_lds_row(Ct, pix_b(integral_constant<int, r - 1>()) * kRowBytes + hl16);
    }
    step2<r, 8 + r>(pa, ca, pb, cb, pen, X, XB);
    if (gen) {
      pa = sel_row(((bia >> r) & 1u) != 0u, ca, pa);
      pb = sel_row(((bib >> r) & 1u) != 0u, cb, pb);
    }
    // (pieces a chain does not have get a defined value too)
    if constexpr (r == 0) Xa[0] = pa;
    if constexpr (NAA < kT ? r == NAA : r == 0) Xa[1] = pa;
    if constexpr (NBA < kT ? r == NBA : r == 0) Xa[2] = pa;
    if constexpr (r == 0) Xb[0] = pb;
    if constexpr (NAB < kT ? r == NAB : r == 0) Xb[1] = pb;
    if constexpr (NBB < kT ? r == NBB : r == 0) Xb[2] = pb;
    ca = can;
    cb = cbn;
  };
  both(integral_constant<int, 7>()); both(integral_constant<int, 6>());
  both(integral_constant<int, 5>()); both(integral_constant<int, 4>());
  both(integral_constant<int, 3>()); both(integral_constant<int, 2>());
  both(integral_constant<int, 1>()); both(integral_constant<int, 0>());
}

// ---- progress counters --------------------------------------------------------------------------
// Waits (one wave) until *p >= need.  Bounded like the sweeps' waits: ~2 s, then the status word is
// set and nobody waits again.
__device__ __forceinline__ unsigned poll_counter(const unsigned *p) {
#ifdef STM_SKEW_X_POLL_RMW
  unsigned v = 0;
  if ((threadIdx.x & 63) == 0) v = __hip_atomic_fetch_add(const_cast<unsigned *>(p), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
#else
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// `seen`: the counter's value when this wave last looked (it only grows): no load at all while
// that already suffices -- a band's owner that runs well behind its neighbour polls once per
// several tiles.
__device__ __forceinline__ void wait_progress(const unsigned *p, unsigned need, unsigned &seen, unsigned *status,
                                              bool &dead) {
#ifdef STM_SKEW_NOWAIT  // (timing experiment: results are wrong)
  return;
#endif
  if (dead || seen >= need) return;
  seen = poll_counter(p);
  if (seen >= need) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (unsigned trip = 1;; ++trip) {
    __builtin_amdgcn_s_sleep(STM_SKEW_POLL_SLEEP);
    seen = poll_counter(p);
    if (seen >= need) return;
    // (the status word is ONE line for the whole chip: looked at every 256th trip only)
    if ((trip & 255u) != 0u) continue;
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { dead = true; return; }
    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
      dead = true;
      if ((threadIdx.x & 63) == 0) atomicOr(status, STM_SGM_STATUS_TIMEOUT);
      return;
    }
  }
}
__device__ __forceinline__ void publish_progress(unsigned *p, unsigned value) {
  __hip_atomic_store(p, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Winner takes all of the finished sum tile (tile_wta_lds with the sheared pixel positions).
__device__ __forceinline__ void skew_wta(const float *St, int wave, int lane, const SkewGeom &G, int x0, int y0,
                                         int32_t *__restrict__ disp, bool sp, int true_paths) {
  constexpr int LP = kWaves, NJ = kD / (LP * 4);  // lanes per pixel: a wave takes 64 / kWaves pixels
  typedef float v4f __attribute__((ext_vector_type(4)));
  // (lane r of a pixel's LP lanes reads the 16-byte chunks r, r + LP, ...: with the sum tile's pixel
  // stride of D + 16 floats the sixteen lanes of each ds_read_b128 lane group -- which are not
  // contiguous, MI355X guide, LDS -- touch sixteen different 16-byte slots, as tile_wta_lds)
  const int pix = wave * (64 / LP) + lane / LP, r = lane % LP;
  const float *p = St + pix * st_stride(kD) + r * 4;
  float best = __builtin_inff();
  int idxk = 0;
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
  m = vmin_f32(m, dpp_f32<0xb1>(m, m));
  m = vmin_f32(m, dpp_f32<0x4e>(m, m));
  if (LP == 8) m = vmin_f32(m, dpp_f32<0x141>(m, m));  // row_half_mirror
  int k = best == m ? idxk + r * 4 : 0x7fffffff;
  k = nan0 ? 0 : k;
  int t;
  t = dpp_i32<0xb1>(k, k); k = t < k ? t : k;
  t = dpp_i32<0x4e>(k, k); k = t < k ? t : k;
  if (LP == 8) { t = dpp_i32<0x141>(k, k); k = t < k ? t : k; }
  const int u = pix & (kT - 1), rr = pix >> 3;
  const int x = x0 + u - rr, y = y0 + rr;
  // (a deferred pixel's winner comes from finish_deferred)
  if (r == 0 && in_image(G, x, y) && !(sp && defer_entry(G, true_paths, x, y) >= 0)) disp[y * G.W + x] = k;
}

// The deferred pixels of a band (stm_sgm_skew_index.h): S = (((S1 + [I2 x pre]) + F2) + I2 x inv2) + F3)
// + I3 x inv3 with the reference's counts, one wave per pixel, then the finished sum and / or its
// winner.  Run by the band's owner behind the band's last publication: off the bands' pipeline.
template <int FINAL>
__device__ __forceinline__ void finish_deferred(const SkewGeom &G, int true_paths, int band, BufRsrc wsb,
                                                float *__restrict__ out, int32_t *__restrict__ disp, int wave,
                                                int lane) {
  constexpr int VPL = kD / 64;
  if (!defer_line_a(G, true_paths)) return;  // (no deferred lines: square images, corrected paths)
  const int d0 = lane * VPL;
  for (int e = wave; e < kDeferPix; e += kWaves) {
    int x, y;
    if (!defer_pixel(G, true_paths, band, e, x, y)) continue;
    Row<VPL> R[kDeferRows];
#pragma unroll
    for (int k = 0; k < kDeferRows; ++k) {
      const unsigned o = (unsigned)(defer_in_img(G, band, e, k) * 4) + (unsigned)d0 * 4u;
#if STM_SKEW_D == 256
      R[k] = buf_load_row4<kBufSc1>(wsb, o, 0);
#else
      typedef unsigned v2u __attribute__((ext_vector_type(2)));
      const v2u q = __builtin_amdgcn_raw_buffer_load_b64(wsb, (int)o, 0, kBufSc1);
      const unsigned q0 = q.x, q1 = q.y;
      R[k].v[0] = __builtin_bit_cast(float, q0); R[k].v[1] = __builtin_bit_cast(float, q1);
#endif
    }
    const tiled::Counts c2 = line_counts(2, x, y, G.H, G.W, true_paths), c3 = line_counts(3, x, y, G.H, G.W, true_paths);
    Row<VPL> S = R[0];
    for (int n = 0; n < c2.pre; ++n) {
#pragma unroll
      for (int k = 0; k < VPL; ++k) S.v[k] = S.v[k] + R[2].v[k];
    }
#pragma unroll
    for (int k = 0; k < VPL; ++k) S.v[k] = S.v[k] + R[1].v[k];
    for (int n = 0; n < c2.inv; ++n) {
#pragma unroll
      for (int k = 0; k < VPL; ++k) S.v[k] = S.v[k] + R[2].v[k];
    }
    for (int n = 0; n < c3.pre; ++n) {
#pragma unroll
      for (int k = 0; k < VPL; ++k) S.v[k] = S.v[k] + R[4].v[k];
    }
#pragma unroll
    for (int k = 0; k < VPL; ++k) S.v[k] = S.v[k] + R[3].v[k];
    for (int n = 0; n < c3.inv; ++n) {
#pragma unroll
      for (int k = 0; k < VPL; ++k) S.v[k] = S.v[k] + R[4].v[k];
    }
    const size_t pix = (size_t)y * G.W + x;
    if constexpr (FINAL != 2) store_row<VPL, true>(out, pix, kD, d0, S);
    if constexpr (FINAL != 0) {
      // Winner takes all (src/winners_take_all.cu:21-40): the first minimum of a strict '<' scan; a
      // NaN never wins unless it is element 0.
      float best = __builtin_inff();
      int idx = 0;
#pragma unroll
      for (int k = 0; k < VPL; ++k) {
        const bool lt = S.v[k] < best;
        best = lt ? S.v[k] : best;
        idx = lt ? d0 + k : idx;
      }
      const float m = wave_min_f32(best);
      int kk = best == m ? idx : 0x7fffffff;
      kk = wave_min_i32(kk);
      const float first = read_lane(S.v[0], 0);
      if (first != first) kk = 0;
      if (lane == 0) disp[pix] = kk;
    }
  }
}

// =====================================================================================
// Forward pass.
// =====================================================================================
struct FwdCarry {
  Row<4> c0, c1, c2a, c2b;  // pair 0; pair 1 (A -> B); pair 2 (P0 -> P1, P1 -> P2)
};
// Entries of the NEXT tile requested during a tile: pair 0's inverse entry, pair 1's three.
struct FwdPre {
  Pre cost;
  Row<4> e0i, e1t, e1i[2];
};

template <int FINAL, bool GEN>
__device__ __forceinline__ void fwd_tile(const SkewArgs &P, const TileCtx &T, const TileCtx &Tn, bool have_next,
                                         const unsigned *Kt, FwdCarry &C,
                                         FwdPre &N, const float *image, float *Ct, float *St, const float *Ip,
                                         Ctx &X, const LaneId &L, int lane, unsigned *publish) {
  const SkewGeom &G = P.G;
  const int w = L.w;
  const int cls2 = count_class(2, G, T.i, T.j, P.true_paths);
  const int cls3 = count_class(3, G, T.i, T.j, P.true_paths);
  const bool sp = cls2 == 3 || cls3 == 3;  // the tile holds pixels of a deferred line
  // Entries of pairs 2 and 3 (a global load takes longer than a pair: they are requested a pair
  // or two ahead; STM_SKEW_FWD_ENTRIES_AT: at the top of the tile, or pair 2's behind pair 0 and
  // pair 3's behind pair 1 -- fewer registers held through the first pairs).
  Row<4> e2t, e3t, Ei2[3], Ei3[3];
  auto load_e2 = [&] {
    const PK K2 = kt_fetch<2>(Kt, L);
    e2t = buf_load_row4<kXSc1>(T.wsb, rowb_off(G, T, 1, T.i - 1, K2.ft, false), 0);
#pragma unroll
    for (int q = 0; q < 3; ++q) Ei2[q] = buf_load_row4<0>(T.wsb, fwd_ie_off(G, T, 4, K2.ie[q]), 0);
  };
  auto load_e3 = [&] {
    const PK K3 = kt_fetch<3>(Kt, L);
    e3t = buf_load_row4<kXSc1>(T.wsb, rowb_off(G, T, 2, T.i - 1, K3.ft, false), 0);
    Ei3[0] = buf_load_row4<0>(T.wsb, fwd_ie_off(G, T, 5, K3.ie[0]), 0);
    Ei3[1] = Ei3[0]; Ei3[2] = Ei3[0];
  };
  if (STM_SKEW_FWD_ENTRIES_AT == 0) { load_e2(); load_e3(); }
  float pen0, pen1, pen2, pen3;
  {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u o = *reinterpret_cast<const v4u *>(Kt + kKtWords + 4 * threadIdx.x);  // (fwd_penalty_table)
    pen0 = penalty_at(Ip, o.x, P.p1, P.p2);
    pen1 = penalty_at(Ip, o.y, P.p1, P.p2);
    pen2 = penalty_at(Ip, o.z, P.p1, P.p2);
    pen3 = penalty_at(Ip, o.w, P.p1, P.p2);
  }
  Row<4> Ef[3], Ei[3], Xf[3];
  // ---- pair 0: right / left
  Ef[0] = C.c0; Ef[1] = C.c0; Ef[2] = C.c0;
  Ei[0] = N.e0i; Ei[1] = N.e0i; Ei[2] = N.e0i;
  STM_SK_STAMP(2);
  fwd_pair_w<0, FINAL, GEN>(w, kt_fetch<0>(Kt, L), Ef, Ei, Xf, 1, false, G, T, P.true_paths, Ct, St, pen0, X, L, [](auto) {});
  STM_SK_STAMP(3);
  C.c0 = Xf[0];
  if (STM_SKEW_FWD_ENTRIES_AT == 1) load_e2();
  // ---- pair 1: down / up
  const PK K1 = kt_fetch<1>(Kt, L);
  Ef[0] = sel_row(K1.carry_first, C.c1, N.e1t);
  Ef[1] = sel_row(K1.carry_first, N.e1t, C.c1);
  Ei[0] = N.e1i[0]; Ei[1] = N.e1i[1]; Ei[2] = N.e1i[1];
  fwd_pair_w<1, FINAL, GEN>(w, K1, Ef, Ei, Xf, 1, false, G, T, P.true_paths, Ct, St, pen1, X, L, [&](auto s_) {
    if constexpr (decltype(s_)::value != kT / 2) return;
    STM_SK_STAMP(4);
#ifndef STM_SKEW_X_NODRAIN
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores of the previous tile have landed
#endif
    STM_SK_STAMP(5);
    lds_barrier();
    STM_SK_STAMP(6);
    if (publish && threadIdx.x == 0) publish_progress(publish, (unsigned)T.j);
  });
  STM_SK_STAMP(7);
  {
    const PK K1b = kt_fetch<1>(Kt, L);
    const bool main_diag = !K1b.restart1;
    C.c1 = sel_row(K1b.carry_first, Xf[1], Xf[0]);
    const Row<4> bottom = sel_row(K1b.carry_first && !main_diag, Xf[0], Xf[1]);
    STM_X_STORE(buf_store_row4<kXSc1>(T.wsb, rowb_off(G, T, 0, T.i, K1b.fb, true), 0, bottom));
  }
  // The next tile's cost rows and its first entries.
  // STM_SKEW_FWD_SPREAD: one request (a cost row, an entry) in front of each step -- 1: of pair 3;
  // 2: of pair 2's second and pair 3's first half; 0: all of them at once (STM_SKEW_FWD_PREFETCH_AT).
  auto prefetch_entry = [&](auto k_) {
    constexpr int k = decltype(k_)::value;
    if constexpr (k == 0) N.e0i = buf_load_row4<0>(Tn.wsb, fwd_ie_off(G, Tn, 0, kt_fetch<0>(Kt, L).ie[0]), 0);
    if constexpr (k == 1) N.e1t = buf_load_row4<kXSc1>(Tn.wsb, have_next ? rowb_off(G, Tn, 0, Tn.i - 1, kt_fetch<1>(Kt, L).ft, false) : kNowhere, 0);
    if constexpr (k == 2) N.e1i[0] = buf_load_row4<0>(Tn.wsb, fwd_ie_off(G, Tn, 3, kt_fetch<1>(Kt, L).ie[0]), 0);
    if constexpr (k == 3) N.e1i[1] = buf_load_row4<0>(Tn.wsb, fwd_ie_off(G, Tn, 3, kt_fetch<1>(Kt, L).ie[1]), 0);
  };
  auto prefetch_next = [&] {
    prefetch_cost(N.cost, G, Tn, image, L);
    prefetch_entry(integral_constant<int, 0>()); prefetch_entry(integral_constant<int, 1>());
    prefetch_entry(integral_constant<int, 2>()); prefetch_entry(integral_constant<int, 3>());
  };
  // (k: 0..7 the cost rows with the four entries beside rows 0..3, the intensities beside row 4)
  auto prefetch_part = [&](auto k_) {
    constexpr int k = decltype(k_)::value;
    prefetch_cost_row<k>(N.cost, G, Tn, L);
    if constexpr (k < 4) prefetch_entry(k_);
    if constexpr (k == 4) prefetch_patch(N.cost, G, Tn, image);
  };
  if (STM_SKEW_FWD_SPREAD == 0 && STM_SKEW_FWD_PREFETCH_AT == 1) prefetch_next();
  if (STM_SKEW_FWD_ENTRIES_AT == 1) load_e3();
  // ---- pair 2: down-right / up-left
  Ef[0] = e2t; Ef[1] = C.c2a; Ef[2] = C.c2b;
  fwd_pair_w<2, FINAL, GEN>(w, kt_fetch<2>(Kt, L), Ef, Ei2, Xf, cls2, sp, G, T, P.true_paths, Ct, St, pen2, X, L, [&](auto s_) {
    constexpr int s = decltype(s_)::value;
    if constexpr (s == kT / 2) { STM_SK_STAMP(8); lds_barrier(); STM_SK_STAMP(9); }
    if constexpr (STM_SKEW_FWD_SPREAD == 2 && s >= kT / 2) prefetch_part(integral_constant<int, s - kT / 2>());
  });
  STM_SK_STAMP(10);
  C.c2a = Xf[0];
  C.c2b = Xf[1];  // (wave 3 has no third piece: unused there)
  // (one unconditional store: a store under a condition is a branch, behind which the compiler
  // can no longer count the accesses in flight -- DESIGN.md 4.1d)
  STM_X_STORE(buf_store_row4<kXSc1>(T.wsb, rowb_off(G, T, 1, T.i, kt_fetch<2>(Kt, L).fb, true), 0, sel_row(nb_of(2, w) < kT, Xf[2], Xf[1])));
  if (STM_SKEW_FWD_SPREAD == 0 && STM_SKEW_FWD_PREFETCH_AT == 2) prefetch_next();
  // ---- pair 3: down-left / up-right
  Ef[0] = e3t; Ef[1] = e3t; Ef[2] = e3t;
  fwd_pair_w<3, FINAL, GEN>(w, kt_fetch<3>(Kt, L), Ef, Ei3, Xf, cls3, sp, G, T, P.true_paths, Ct, St, pen3, X, L, [&](auto s_) {
    constexpr int s = decltype(s_)::value;
    if constexpr (s == kT / 2) { STM_SK_STAMP(11); lds_barrier(); STM_SK_STAMP(12); }
    if constexpr (STM_SKEW_FWD_SPREAD == 1) prefetch_part(s_);
    if constexpr (STM_SKEW_FWD_SPREAD == 2 && s < kT / 2) prefetch_part(integral_constant<int, s + kT / 2>());
  });
  STM_SK_STAMP(13);
  STM_X_STORE(buf_store_row4<kXSc1>(T.wsb, rowb_off(G, T, 2, T.i, kt_fetch<3>(Kt, L).fb, true), 0, Xf[0]));
}

__device__ __forceinline__ void make_tile(TileCtx &T, const SkewGeom &G, int i, int j) {
  make_tile_pos(T, G, i, j);
#ifdef STM_SKEW_X_NOGEN  // (timing experiment: every tile takes the fast form; results are wrong)
  T.gen = false;
#endif
}

template <int FINAL>
__global__ __launch_bounds__(kThreads, kD == 128 ? 2 : 1) void sgm_skew_fwd_kernel(const float *__restrict__ cost_all,
                                                                  const float *__restrict__ image_all,
                                                                  float *__restrict__ out_all,
                                                                  int32_t *__restrict__ disp_all,
                                                                  float *__restrict__ ws, const SkewArgs P) {
  __shared__ __attribute__((aligned(16))) float Ct[kT * kT * kD];
  __shared__ __attribute__((aligned(16))) float St[kT * kT * st_stride(kD)];
  __shared__ float Ip[kIpN];
  __shared__ __attribute__((aligned(16))) unsigned Kt[kKtWords + 4 * kThreads];
  __shared__ int s_unit;
  const SkewGeom &G = P.G;
  const int lane = threadIdx.x & 63;
  LaneId L;
  L.init();
  Ctx X;
  ctx_init(X, P.p1, L, P.dlim);
  kt_fill<0>(Kt, L); kt_fill<1>(Kt, L); kt_fill<2>(Kt, L); kt_fill<3>(Kt, L);  // (read behind the loop's first barrier)
  {
    // fwd_penalty_table: behind the chain table, four words per lane -- where the lane's samples of
    // each pair's penalties lie in the intensity patch (chain_penalty_offsets).
    PK K0, K1, K2, K3;
    pk_init<0>(K0, L); pk_init<1>(K1, L); pk_init<2>(K2, L); pk_init<3>(K3, L);
    const int w = L.w;
    unsigned *po = Kt + kKtWords + 4 * threadIdx.x;
    po[0] = chain_penalty_offsets<0, -1, 0, 1>(K0, kT, kT, K0, kT, kT, lane);
    po[1] = chain_penalty_offsets<1, -1, 1, 1>(K1, na_of(1, w), kT, K1, na_of(1, w), kT, lane);
    po[2] = chain_penalty_offsets<2, -1, 2, 1>(K2, na_of(2, w), nb_of(2, w), K2, na_of(2, w), nb_of(2, w), lane);
    po[3] = chain_penalty_offsets<3, -1, 3, 1>(K3, kT, kT, K3, kT, kT, lane);
  }
  unsigned *status = reinterpret_cast<unsigned *>(ws);
  unsigned *prog = reinterpret_cast<unsigned *>(ws + G.prog_off);
  const size_t npix = (size_t)G.H * G.W;
  const int nunits = G.batch * G.nTR;
  bool dead = false;
  FwdCarry C;
  FwdPre N;
  for (;;) {
    __syncthreads();  // (the previous unit's last readers of s_unit, Ct, St)
    if (threadIdx.x == 0) s_unit = (int)atomicAdd(status + kTicketFwd, 1u);
    __syncthreads();
    const int unit = s_unit;
    if (unit >= nunits) break;
    const int b = unit % G.batch, i = unit / G.batch;
    const float *image = image_all + (size_t)b * npix;
    int32_t *disp = disp_all ? disp_all + (size_t)b * npix : nullptr;
    TileCtx T, Tn;
    T.cost = make_rsrc(cost_all + (size_t)b * npix * kD, (unsigned)(npix * kD * 4));
    T.out = make_rsrc(out_all ? out_all + (size_t)b * npix * kD : ws, out_all ? (unsigned)(npix * kD * 4) : 0u);
    T.wsb = make_rsrc(ws + img_base(G, b), (unsigned)(G.img_floats * 4));
    Tn = T;
    const unsigned *prog_above = prog + prog_index(G, 1, b, i > 0 ? i - 1 : 0);
    unsigned *prog_mine = prog + prog_index(G, 1, b, i);
    // Tile 0: everything that later tiles get from their predecessor's prefetch.
    make_tile(T, G, i, 0);
    unsigned seen = 0, ahead = 0;
    if (L.w == 0 && i > 0) wait_progress(prog_above, (unsigned)progress_needed(G, 0), seen, status, dead);
    __syncthreads();
    prefetch_cost(N.cost, G, T, image, L);
    {
      const PK K0 = kt_fetch<0>(Kt, L), K1 = kt_fetch<1>(Kt, L);
      N.e0i = buf_load_row4<0>(T.wsb, fwd_ie_off(G, T, 0, K0.ie[0]), 0);
      N.e1t = buf_load_row4<kXSc1>(T.wsb, rowb_off(G, T, 0, i - 1, K1.ft, false), 0);
      N.e1i[0] = buf_load_row4<0>(T.wsb, fwd_ie_off(G, T, 3, K1.ie[0]), 0);
      N.e1i[1] = buf_load_row4<0>(T.wsb, fwd_ie_off(G, T, 3, K1.ie[1]), 0);
    }
    C.c0 = N.e0i; C.c1 = N.e0i; C.c2a = N.e0i; C.c2b = N.e0i;  // (any defined value: tile 0's paths begin in it)
    for (int j = 0; j < G.nTC; ++j) {
      make_tile(T, G, i, j);
#ifdef STM_SKEW_STAMPS
      T.stamp = P.stamps && b == 0 && i == P.stamp_band && j < 64;
#endif
      STM_SK_STAMP(0);
      const bool have_next = j + 1 < G.nTC;
      make_tile(Tn, G, i, have_next ? j + 1 : j);
      stage_cost(N.cost, Ct, Ip, L);
      // The band above is far enough ahead for this tile's entries and the next tile's prefetch.
      // `ahead`: the counter as requested a tile ago -- a band that follows its neighbour closely
      // needs a new value every tile, and a poll at this point is a memory round trip on the
      // band's critical path: its tiles then take that much longer than its neighbour's, it falls
      // behind until the polls become rare, and so does every band behind it, each by a little
      // more (a single 1080p image: 40 ms instead of 13).  Requested early, the value is here.
      if (L.w == 0 && i > 0) {
        seen = ahead > seen ? ahead : seen;
        wait_progress(prog_above, (unsigned)progress_needed(G, j), seen, status, dead);
      }
      STM_SK_STAMP(1);
      lds_barrier();
      if (L.w == 0 && i > 0) ahead = poll_counter(prog_above);  // (looked at at the top of the next tile)
      // (the band's own progress -- tiles < j are complete -- is published behind pair 1's barrier,
      // in front of which every wave drains its stores: by then the previous tile's last stores
      // have had a pair's time, a drain at the top of the tile would wait for them)
      unsigned *publish = j > 0 ? prog_mine : nullptr;
      if (T.gen) fwd_tile<FINAL, true>(P, T, Tn, have_next, Kt, C, N, image, Ct, St, Ip, X, L, lane, publish);
      else fwd_tile<FINAL, false>(P, T, Tn, have_next, Kt, C, N, image, Ct, St, Ip, X, L, lane, publish);
      lds_barrier();
      STM_SK_STAMP(14);
      if constexpr (FINAL != 0) {
        const bool sp = count_class(2, G, i, j, P.true_paths) == 3 || count_class(3, G, i, j, P.true_paths) == 3;
        skew_wta(St, L.w, lane, G, T.x0, T.y0, disp, sp, P.true_paths);
      }
      STM_SK_STAMP(15);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) publish_progress(prog_mine, (unsigned)G.nTC);
    // (behind the barrier every wave's stores of the band's deferred pixels have landed)
    finish_deferred<FINAL>(G, P.true_paths, i, T.wsb, out_all ? out_all + (size_t)b * npix * kD : nullptr, disp, L.w, lane);
  }
}

// =====================================================================================
// Reverse pass.
// =====================================================================================
struct RevCarry {
  Row<4> c0, c1, c2a, c2b;  // pair 0; pair 1 (B -> A); pair 2 (P1 -> P0, P2 -> P1)
};

// Bottom entries of a tile (the band below wrote them in this launch: sc1), requested a tile ahead.
struct RevPre {
  Pre cost;
  Row<4> b1, b2, b3;
};
__device__ __forceinline__ void rev_bottoms(RevPre &N, const SkewGeom &G, const TileCtx &T, const PK &K1, const PK &K2,
                                            const PK &K3) {
  N.b1 = buf_load_row4<kXSc1>(T.wsb, rowb_off(G, T, 3, T.i + 1, K1.rb, false), 0);
  N.b2 = buf_load_row4<kXSc1>(T.wsb, rowb_off(G, T, 4, T.i + 1, K2.rb, false), 0);
  N.b3 = buf_load_row4<kXSc1>(T.wsb, rowb_off(G, T, 5, T.i + 1, K3.rb, false), 0);
}

__device__ __forceinline__ void rev_tile(const SkewArgs &P, const TileCtx &T, const TileCtx &Tn, const PK &K0,
                                         const PK &K1, const PK &K2, const PK &K3, RevCarry &C, RevPre &N,
                                         const float *image, const float *Ct, const float *Ip, Ctx &X,
                                         const LaneId &L, int lane, unsigned *publish, unsigned published,
                                         unsigned po01, unsigned po23) {
  const SkewGeom &G = P.G;
  const int w = L.w;
  const Row<4> b1 = N.b1, b2 = N.b2, b3 = N.b3;
  const float pen01 = penalty_at(Ip, po01, P.p1, P.p2);
  const float pen23 = penalty_at(Ip, po23, P.p1, P.p2);
  Row<4> Ea[3], Eb[3], Xa[3], Xb[3];
  // ---- pairs 0 (left) and 1 (up)
  Ea[0] = C.c0; Ea[1] = C.c0; Ea[2] = C.c0;
  // pair 1: the piece that enters from the right side takes the carry, the other one the bottom state
  Eb[0] = sel_row(K1.e0_bottom, b1, C.c1);
  Eb[1] = sel_row(K1.carry_first, C.c1, b1);
  Eb[2] = b1;
#define STM_REV01(NA) rev_pairs<0, kT, kT, 1, NA, kT>(K0, Ea, Xa, K1, Eb, Xb, G, T, Ct, pen01, X, L)
  if constexpr (kD == 128) {
    switch (w) {
      case 0: STM_REV01(7); break;
      case 1: STM_REV01(6); break;
      case 2: STM_REV01(5); break;
      default: STM_REV01(4); break;
    }
  } else {
    switch (w) {
      case 0: STM_REV01(kT); break;
      case 1: STM_REV01(7); break;
      case 2: STM_REV01(6); break;
      case 3: STM_REV01(5); break;
      case 4: STM_REV01(4); break;
      case 5: STM_REV01(3); break;
      case 6: STM_REV01(2); break;
      default: STM_REV01(1); break;
    }
  }
#undef STM_REV01
  // This wave's stores of the PREVIOUS tile have landed (they have had half a tile's time; a drain
  // at the top of the tile would wait for them), the workgroup's barrier, then the band's progress
  // -- the tiles before this one are complete -- is published.
#ifndef STM_SKEW_X_NODRAIN
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  lds_barrier();
  if (publish && threadIdx.x == 0) publish_progress(publish, published);
  C.c0 = Xa[0];
  STM_X_STORE(buf_store_row4<STM_SKEW_SIDE_AUX>(T.wsb, side_off(G, T, T.j, K0.rs[0]), 0, Xa[0]));
  {
    // slot order [q0, q1]; the inverse chain leaves q1 at slot NA (Xb[1]) and q0 at slot 0 (Xb[0]).
    // first half-wave: q1 = B leaves to the left (carry + side), q0 = A leaves at the top;
    // second: q1 = A' leaves at the top, q0 = B' to the left; main diagonal: leaves at the top (Xb[0]).
    const bool main_diag = !K1.restart1;
    const Row<4> left = sel_row(K1.carry_first, Xb[0], Xb[1]);
    const Row<4> top = sel_row(K1.carry_first && !main_diag, Xb[1], Xb[0]);
    C.c1 = left;
    const unsigned so = K1.carry_first ? K1.rs[0] : K1.rs[1];
    STM_X_STORE(buf_store_row4<STM_SKEW_SIDE_AUX>(T.wsb, side_off(G, T, T.j, so), 0, left));
    STM_X_STORE(buf_store_row4<kXSc1>(T.wsb, rowb_off(G, T, 3, T.i, K1.rt, false), 0, top));
  }
  // The next tile's cost rows and bottom entries (half a tile ahead; held in registers no longer -- a whole
  // tile ahead measured the same).
  prefetch_cost(N.cost, G, Tn, image, L);
  rev_bottoms(N, G, Tn, K1, K2, K3);
  // ---- pairs 2 (up-left) and 3 (up-right)
  Eb[0] = b3; Eb[1] = b3; Eb[2] = b3;
  const bool three = nb_of(2, w) < kT;  // the chain has a third piece
  Ea[0] = C.c2a; Ea[1] = sel_row(three, C.c2b, b2); Ea[2] = b2;
#define STM_REV23(NA, NB) rev_pairs<2, NA, NB, 3, kT, kT>(K2, Ea, Xa, K3, Eb, Xb, G, T, Ct, pen23, X, L)
  switch (kD == 128 ? w : w >> 1) {
    case 0: STM_REV23(1, 5); break;
    case 1: STM_REV23(2, 6); break;
    case 2: STM_REV23(3, 7); break;
    default: STM_REV23(4, kT); break;
  }
#undef STM_REV23
  // pair 2: P0 leaves at the top; P1 to the left (-> the next tile's P0), P2 to the left (-> its P1)
  STM_X_STORE(buf_store_row4<kXSc1>(T.wsb, rowb_off(G, T, 4, T.i, K2.rt, false), 0, Xa[0]));
  C.c2a = Xa[1];
  STM_X_STORE(buf_store_row4<STM_SKEW_SIDE_AUX>(T.wsb, side_off(G, T, T.j, K2.rs[1]), 0, Xa[1]));
  C.c2b = sel_row(three, Xa[2], Xa[1]);
  STM_X_STORE(buf_store_row4<STM_SKEW_SIDE_AUX>(T.wsb, side_off(G, T, T.j, K2.rs[2]), 0, C.c2b));  // (wave 3: no third piece, offset kNowhere)
  STM_X_STORE(buf_store_row4<kXSc1>(T.wsb, rowb_off(G, T, 5, T.i, K3.rt, false), 0, Xb[0]));
}

__global__ __launch_bounds__(kThreads, kD == 128 ? STM_SKEW_REV_OCC : 1) void sgm_skew_rev_kernel(const float *__restrict__ cost_all,
                                                                  const float *__restrict__ image_all,
                                                                  float *__restrict__ ws, const SkewArgs P) {
  __shared__ __attribute__((aligned(16))) float Ct[kT * kT * kD];
  __shared__ float Ip[kIpN];
  __shared__ int s_unit;
  const SkewGeom &G = P.G;
  const int lane = threadIdx.x & 63;
  LaneId L;
  L.init();
  Ctx X;
  ctx_init(X, P.p1, L, P.dlim);
  PK K0, K1, K2, K3;
  pk_init<0>(K0, L); pk_init<1>(K1, L); pk_init<2>(K2, L); pk_init<3>(K3, L);
  // (where the lane's samples of the two penalty sets lie in the intensity patch: the same for every tile)
  const unsigned po01 = chain_penalty_offsets<0, 1, 1, 1>(K0, kT, kT, K1, na_of(1, L.w), kT, lane);
  const unsigned po23 = chain_penalty_offsets<2, 1, 3, 1>(K2, na_of(2, L.w), nb_of(2, L.w), K3, kT, kT, lane);
  unsigned *status = reinterpret_cast<unsigned *>(ws);
  unsigned *prog = reinterpret_cast<unsigned *>(ws + G.prog_off);
  const size_t npix = (size_t)G.H * G.W;
  const int nunits = G.batch * G.nTR;
  bool dead = false;
  RevCarry C;
  RevPre N;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) s_unit = (int)atomicAdd(status + kTicketRev, 1u);
    __syncthreads();
    const int unit = s_unit;
    if (unit >= nunits) break;
    const int b = unit % G.batch, i = G.nTR - 1 - unit / G.batch;
    const float *image = image_all + (size_t)b * npix;
    TileCtx T, Tn;
    T.cost = make_rsrc(cost_all + (size_t)b * npix * kD, (unsigned)(npix * kD * 4));
    T.out = T.cost;
    T.wsb = make_rsrc(ws + img_base(G, b), (unsigned)(G.img_floats * 4));
    Tn = T;
    const bool below = i + 1 < G.nTR;
    const unsigned *prog_below = prog + prog_index(G, 0, b, below ? i + 1 : i);
    unsigned *prog_mine = prog + prog_index(G, 0, b, i);
    make_tile(T, G, i, G.nTC - 1);
    unsigned seen = 0, ahead = 0;
    if (L.w == 0 && below) wait_progress(prog_below, (unsigned)progress_needed(G, 0), seen, status, dead);
    __syncthreads();
    prefetch_cost(N.cost, G, T, image, L);
    rev_bottoms(N, G, T, K1, K2, K3);
    Row<4> zero;
#pragma unroll
    for (int k = 0; k < 4; ++k) zero.v[k] = 0.0f;
    C.c0 = zero; C.c1 = zero; C.c2a = zero; C.c2b = zero;  // (the last tile's paths begin in it)
    for (int jj = 0; jj < G.nTC; ++jj) {
      const int j = G.nTC - 1 - jj;
      make_tile(T, G, i, j);
      make_tile(Tn, G, i, j > 0 ? j - 1 : 0);
      stage_cost(N.cost, Ct, Ip, L);
      // The band below is far enough ahead for the next tile's prefetch (`ahead`: see the forward pass).
      if (L.w == 0 && below) {
        seen = ahead > seen ? ahead : seen;
        wait_progress(prog_below, (unsigned)progress_needed(G, jj), seen, status, dead);
      }
      lds_barrier();
      if (L.w == 0 && below) ahead = poll_counter(prog_below);
      unsigned *publish = jj > 0 ? prog_mine : nullptr;
      rev_tile(P, T, Tn, K0, K1, K2, K3, C, N, image, Ct, Ip, X, L, lane, publish, (unsigned)jj, po01, po23);
      lds_barrier();  // (the next tile's staging overwrites the cost tile)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) publish_progress(prog_mine, (unsigned)G.nTC);
  }
}

}  // namespace

// One call's two launches (stm_sgm_skew_launch.h).
#if STM_SKEW_D == 128
int skew_launch_d128
#else
int skew_launch_d256
#endif
    (const float *cost, const float *left, float *out, int32_t *disp, float *ws, const skew::SkewGeom &G, float p1,
     float p2, int dlim, int true_paths, hipStream_t stream, int final_mode, int cus) {
  SkewArgs A;
  A.G = G; A.p1 = p1; A.p2 = p2; A.dlim = dlim; A.true_paths = true_paths;
#ifdef STM_SKEW_STAMPS
  A.stamps = g_skew_stamps; A.stamp_band = g_skew_stamp_band;
#endif
  const long long units = (long long)G.batch * G.nTR;
  {
    const long long wgs = (long long)cus * (kD == 128 ? STM_SKEW_REV_OCC : 1);
    const dim3 grid((unsigned)(units < wgs ? units : wgs)), block(kThreads);
    hipLaunchKernelGGL(sgm_skew_rev_kernel, grid, block, 0, stream, cost, left, ws, A);
    const int rc = check_launch("stm_sgm_f32 (skewed tiles: reverse pass)");
    if (rc != STM_OK) return rc;
  }
  const long long wgs = (long long)cus * (kD == 128 ? 2 : 1);
  const dim3 grid((unsigned)(units < wgs ? units : wgs)), block(kThreads);
  if (final_mode == 0)
    hipLaunchKernelGGL((sgm_skew_fwd_kernel<0>), grid, block, 0, stream, cost, left, out, disp, ws, A);
  else if (final_mode == 1)
    hipLaunchKernelGGL((sgm_skew_fwd_kernel<1>), grid, block, 0, stream, cost, left, out, disp, ws, A);
  else
    hipLaunchKernelGGL((sgm_skew_fwd_kernel<2>), grid, block, 0, stream, cost, left, out, disp, ws, A);
  return check_launch("stm_sgm_f32 (skewed tiles: forward pass)");
}

}  // namespace stm

#if defined(STM_SKEW_STAMPS)
#if STM_SKEW_D == 128
extern "C" void stm_debug_set_skew_stamps_d128(void *p, int band) {
#else
extern "C" void stm_debug_set_skew_stamps_d256(void *p, int band) {
#endif
  stm::g_skew_stamps = static_cast<unsigned long long *>(p);
  stm::g_skew_stamp_band = band;
}
#endif
