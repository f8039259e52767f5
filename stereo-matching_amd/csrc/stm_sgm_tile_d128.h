// Pieces of the D = 128 tile kernel (stm_sgm_tile_h2.hip) that do not depend on how a wave steps
// its lines: the lines a wave owns, a tile's terms of the entry addresses, and the
// one-line-per-wave forms that the tiles on the image border take.  (A second kernel on top of
// them -- four pixels per wave, forward | inverse chains in the two wave halves -- was measured
// and is in the history only: DESIGN.md section 9.)
#pragma once

#include "stm_sgm_tile_common.h"

namespace stm {
namespace {

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
  h2_parts(PHASE, w, half, p[0], p[1], K.restart);
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

// A tile's terms of the entry addresses; real = false: the tile is not interior, its entry
// states are not used and every lane reads the same valid dummy row instead (`row`).
struct H2TileBase {
  const float *row, *col;
  bool real;
};
template <typename Walk>
__device__ __forceinline__ H2TileBase h2_tile_base(float *ws, const TiledGeom &G, const Walk &t) {
  H2TileBase T;
  T.row = ws + G.rowb_off + (long long)t.b * G.rowb_img + h2_tile_row_term(G, t.trow, t.tcol);
  T.col = ws + G.colb_off + (long long)t.b * G.colb_img + h2_tile_col_term(G, t.trow, t.tcol);
  T.real = true;
  return T;
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
    Q.g[i] = tile_geom(PHASE, 2 * w + i, tw, th);
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


}  // namespace
}  // namespace stm
