// Index model of the SKEWED-TILE form of the semiglobal aggregation (stm_sgm_skew.hip): the
// geometry of its tiles and workspace and every expression by which its kernels form a global
// address.  Plain C++, host and device: the kernels call these functions, and
// tests/host/skew_index_check.cpp (tests/test_tiled_index.py, no GPU) evaluates them for every
// tile of a grid of shapes and asserts that each access lies inside its buffer.
//
// Why skewed tiles.  The tiled form (stm_sgm_tiled.hip) rebuilds all eight directions of an 8 x 8
// tile from border states, and those border states cost three more reads of the cost volume (two
// row sweeps and a walk) in which every direction is computed a second time.  A tile pass could
// CARRY a direction from tile to tile instead -- but with rectangular tiles only three of the four
// "forward" directions (right, down, down-right, down-left) are causal in one tile order: the
// down-left direction enters a tile from its RIGHT neighbour, the other two from the left one.
// Shearing every 8-row band by one column per row makes all four causal: with
//     tile (i, j), pixel (u, r), u, r in [0, 8):   x = 8 j + u - r,   y = 8 i + r
// the four directions step by (du, dr) = right (1, 0), down (1, 1), down-right (2, 1), down-left
// (0, 1): every one of them leaves a tile to the right or downwards, and a tile depends only on
// the tile to its left (same workgroup: registers) and on tiles j and j + 1 of the band above
// (another workgroup: row border states in memory + a progress counter).  Two passes then do
// everything, each reading the cost volume once:
//   reverse pass  (bands bottom-up, tiles right-to-left): carries left, up, up-left, up-right and
//                 stores their states where a path leaves a tile: top rows (rowb) and left sides
//                 (side blocks);
//   forward pass  (bands top-down, tiles left-to-right): carries right, down, down-right,
//                 down-left, rebuilds the four inverse directions inside the tile from the
//                 reverse pass's states, adds all eight in the reference's order into the
//                 LDS-resident sum tile and reduces / stores it.
// 8 + ~10 bytes per element instead of 30, and every direction computed 1.5 times instead of 2.
//
// All offsets are in floats (4-byte elements) unless a name says otherwise.
#pragma once

#include "stm_sgm_tiled_index.h"  // STM_HD, imin_/imax_, Counts/seg_counts

// Loops over a chain's (at most three) pieces must be unrolled in device code: an array indexed
// by a loop variable lives in scratch memory.
#if defined(__HIPCC__)
#define STM_UNROLL _Pragma("unroll")
#else
#define STM_UNROLL
#endif

namespace stm {
namespace skew {

using tiled::imax_;
using tiled::imin_;

using tiled::kT;                 // tile side (8)
constexpr int kSide = 29;          // states a tile hands to its left / right neighbour (side block)
constexpr int kHeaderFloats = 64;  // 256 bytes: status word, two unit tickets, padding
constexpr int kTicketRev = 1, kTicketFwd = 2;  // 32-bit words of the header
// Deferred pixels (defer_line): per band up to 8 pixels of either re-walked line, five rows each:
// the sum before pair 2, pair 2's forward and inverse values, pair 3's.
constexpr int kDeferPix = 16, kDeferRows = 5;
// A progress counter has a 128-byte line to itself: a band's owner polls its neighbour's counter
// while it waits, and with the counters of 32 bands in one line every publishing store queued up
// behind the polls of up to 32 waiting workgroups (a single 1080p image: 40 ms instead of 13).
constexpr int kProgStride = 32;  // 32-bit words between two counters

// Pairs in the reference's order: 0 right / left, 1 down / up, 2 down-right / up-left,
// 3 down-left / up-right.  Step of a pair's FORWARD direction in the image and in a tile.
STM_HD constexpr int img_dx(int phase) { return phase == 0 ? 1 : (phase == 1 ? 0 : (phase == 2 ? 1 : -1)); }
STM_HD constexpr int img_dy(int phase) { return phase == 0 ? 0 : 1; }
STM_HD constexpr int tile_du(int phase) { return phase == 0 ? 1 : (phase == 1 ? 1 : (phase == 2 ? 2 : 0)); }
STM_HD constexpr int tile_dr(int phase) { return phase == 0 ? 0 : 1; }
// Distance of two consecutive pixels of a line in a tile's pixel index r * 8 + u.
STM_HD constexpr int tile_pstep(int phase) { return tile_dr(phase) * kT + tile_du(phase); }

struct SkewGeom {
  int H, W, D;
  int nTR, nTC;         // bands of 8 rows; skewed tile columns: u = x + r covers [0, W + 7)
  int batch;
  long long prog_off;   // [2 passes][batch][nTR] progress counters (tiles finished in a band), one per 128-byte line
  long long img_off;    // per image: [6][nTR][W][D] row border states, then [nTR][nTC][kSide][D] side blocks
  long long img_floats;
  long long side_in_img;  // offset of the side blocks inside an image's block
  long long defer_in_img; // [nTR][kDeferPix][kDeferRows][D]: pixels of the re-walked diagonals (deferred updates)
};

// Lays the workspace out; returns false when an image's blocks do not fit the 32-bit byte offsets
// of a buffer descriptor.  `zero_floats`: the head (status, tickets, progress counters) that
// every call clears.
STM_HD bool make_geom(int H, int W, int D, int batch, SkewGeom &G, long long &total_floats,
                      long long &zero_floats) {
  G.H = H; G.W = W; G.D = D;
  G.nTR = (H + kT - 1) / kT;
  G.nTC = (W + kT - 1 + kT - 1) / kT;
  G.batch = batch;
  long long off = kHeaderFloats;
  G.prog_off = off;
  off += 2ll * batch * G.nTR * kProgStride;
  zero_floats = off;
  G.img_off = off;
  G.side_in_img = 6ll * G.nTR * W * D;
  G.defer_in_img = G.side_in_img + (long long)G.nTR * G.nTC * kSide * D;
  G.img_floats = G.defer_in_img + (long long)G.nTR * kDeferPix * kDeferRows * D;
  off += G.img_floats * batch;
  total_floats = off;
  const long long lim = 0xfffffff0ll;  // (offsets >= this mark "no access" in the kernels)
  return (long long)H * W * D * 4 < lim && G.img_floats * 4 < lim;
}

// ---- pixels -----------------------------------------------------------------------------------
STM_HD int tile_x(int j, int u, int r) { return kT * j + u - r; }
STM_HD int tile_y(int i, int r) { return kT * i + r; }
STM_HD bool in_image(const SkewGeom &G, int x, int y) { return x >= 0 && x < G.W && y >= 0 && y < G.H; }

// A tile all of whose pixels and all of whose pixels' eight neighbours lie inside the image: no
// path begins in it, every access of its fast form is in range.
STM_HD bool tile_fast(const SkewGeom &G, int i, int j) {
  return i >= 1 && kT * i + kT < G.H && j >= 1 && kT * j + kT < G.W;
}

// ---- row border states --------------------------------------------------------------------------
// Direction index: 0 down, 1 down-right, 2 down-left -- state AFTER the pixel (x, 8 t + 7), the last
// row of band t; 3 up, 4 up-left, 5 up-right -- state after the pixel (x, 8 t), the band's first row.
// Offset inside an image's block.
STM_HD long long rowb_in_img(const SkewGeom &G, int dir, int t, int x) {
  return (((long long)dir * G.nTR + t) * G.W + x) * (long long)G.D;
}
// ---- side blocks --------------------------------------------------------------------------------
// Block (t, j): the states with which the leftward directions LEAVE tile j of band t to the left,
// i.e. after its pixels (0, r) (left: r = 0..7 -> index r; up: r = 1..7 -> 7 + r) and (u, r), u in
// {0, 1}, r = 1..7 (up-left: 15 + 2 (r - 1) + u).  Tile j - 1 of the forward pass enters its
// inverse chains from them.
STM_HD int side_idx_left(int r) { return r; }
STM_HD int side_idx_up(int r) { return 7 + r; }
STM_HD int side_idx_upleft(int u, int r) { return 15 + 2 * (r - 1) + u; }
STM_HD long long side_in_img(const SkewGeom &G, int t, int j, int idx) {
  return G.side_in_img + (((long long)t * G.nTC + j) * kSide + idx) * (long long)G.D;
}
STM_HD long long img_base(const SkewGeom &G, int b) { return G.img_off + (long long)b * G.img_floats; }
STM_HD long long prog_index(const SkewGeom &G, int pass, int b, int t) {
  return (((long long)pass * G.batch + b) * G.nTR + t) * kProgStride;
}

// ---- chains ---------------------------------------------------------------------------------------
// The 64 pixels of a tile are, per pair, the 8 slots of 8 CHAINS.  A chain is made of up to three
// PIECES, each a run of consecutive pixels of one line of the pair's direction inside the tile;
// piece p + 1 lies on the line that piece p's line becomes in the NEXT tile to the right, so the
// state with which the forward direction leaves piece p to the right is the state with which it
// enters piece p + 1 of the same chain one tile later: a chain's owner carries it in registers.
//   pair 0 (rows)            chain k: row r = k, one piece of 8.
//   pair 3 (columns u = k)   one piece of 8, enters at the top, leaves at the bottom.
//   pair 1 (lines u - r = c) chain c = 1..7: A = line c (8 - c pixels from (c, 0), top -> right side),
//                            B = line c - 8 (c pixels from (0, 8 - c), left side -> bottom); chain 0: the
//                            main diagonal (top -> bottom).
//   pair 2 (lines u - 2r = c) chain c0 = 0..7: P0 = line c0 from (c0, 0) (top -> right), P1 = line
//                            c0 - 8 (left -> right, or -> bottom for c0 < 2), P2 = line c0 - 16 (left ->
//                            bottom; c0 >= 2): 1+4+3, 1+4+3, 2+4+2, 2+4+2, 3+4+1, 3+4+1, 4+4, 4+4 pixels.
struct Piece {
  int u, r, n;  // first pixel (forward order) and length
};
struct Chain {
  Piece p[3];
  int np;
};
STM_HD Chain chain_of(int phase, int k) {
  Chain c;
  c.np = 1;
  c.p[1].u = c.p[1].r = c.p[1].n = 0;
  c.p[2] = c.p[1];
  if (phase == 0) {
    c.p[0].u = 0; c.p[0].r = k; c.p[0].n = kT;
  } else if (phase == 3) {
    c.p[0].u = k; c.p[0].r = 0; c.p[0].n = kT;
  } else if (phase == 1) {
    c.p[0].u = k; c.p[0].r = 0; c.p[0].n = kT - k;
    if (k > 0) {
      c.np = 2;
      c.p[1].u = 0; c.p[1].r = kT - k; c.p[1].n = k;
    }
  } else {
    const int n0 = (kT - 1 - k) / 2 + 1;
    c.p[0].u = k; c.p[0].r = 0; c.p[0].n = n0;
    c.np = k >= 2 ? 3 : 2;
    c.p[1].u = k & 1; c.p[1].r = (kT - k + (k & 1)) / 2; c.p[1].n = 4;
    if (k >= 2) {
      c.p[2].u = k & 1; c.p[2].r = (2 * kT - k + (k & 1)) / 2; c.p[2].n = kT - n0 - 4;
    }
  }
  return c;
}

// ---- a lane's chain of a pair: where its pieces lie and what it loads and stores ------------------
// D = 128: two chains per wave (half-waves: lanes 0-31 / 32-63), four waves per tile; D = 256: one
// chain per wave, eight waves.  Pieces are in SLOT order (which for pair 1's second half-wave at
// D = 128 is B before A, so that both halves of a wave change piece at the same slot).
// Relative byte offsets: a row border state at image column x = 8 j + xr is ((xr + 8) * D + d0) * 4
// from the state of column 8 j - 8 in the same line; a side state is ((idx * D + d0) * 4) | 1 into
// its block.  kNoAccess: none (an offset beyond every buffer: the hardware drops the access).
constexpr unsigned kNoAccess = 0xfffffff0u;
struct ChainIO {
  int pix[3];       // r * 8 + u of the first slot of each piece
  unsigned ft;      // forward pass: top entry of the forward chain (row border states; relative)
  unsigned fb;      // forward pass: bottom exit of the forward chain
  unsigned ie[3];   // forward pass: entry of the inverse chain per piece: relative | 1 = side block
  unsigned rb;      // reverse pass: bottom entry of the inverse chain
  unsigned rt;      // reverse pass: top exit of the inverse chain
  unsigned rs[3];   // reverse pass: left exit per piece (side block; kNoAccess: none)
  bool carry_first; // pair 1: the first piece in slot order is the one that enters from the left
  bool restart1;    // the second piece begins at an entry state (false: the main diagonal continues)
  bool e0_bottom;   // pair 1, reverse pass: the first piece in slot order enters from the band below
};
STM_HD unsigned rel_row(int D, int xr, int d0) { return (unsigned)(((xr + kT) * D + d0) * 4); }
STM_HD unsigned rel_side(int D, int idx, int d0) { return (unsigned)((idx * D + d0) * 4) | 1u; }
// The chain of the simple pairs (tile row of pair 0, tile column of pair 3) of wave w / half-wave.
STM_HD int lane_row(int D, int w, int half) { return D == 128 ? 2 * w + half : w; }
// (a template over the pair: with a run-time `phase` hipcc kept the unused branches' values alive
// around the kernels' tile loops -- 111 instead of 19 spilled registers in the forward pass)
template <int PHASE>
STM_HD void chain_io_t(int D, int w, int half, int d0, ChainIO &K) {
  const int row = lane_row(D, w, half);
  K.carry_first = false;
  K.restart1 = true;
  K.e0_bottom = false;
  K.ft = K.fb = K.rb = K.rt = kNoAccess;
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int q = 0; q < 3; ++q) { K.pix[q] = 0; K.ie[q] = kNoAccess; K.rs[q] = kNoAccess; }
  if constexpr (PHASE == 0) {
    const int k = row;
    K.pix[0] = k * kT;
    K.ie[0] = rel_side(D, side_idx_left(k), d0);
    K.rs[0] = rel_side(D, side_idx_left(k), d0);
  } else if constexpr (PHASE == 3) {
    const int k = row;
    K.pix[0] = k;
    K.ft = rel_row(D, k + 1, d0);      // state of down-left after (x + 1, y0 - 1)
    K.fb = rel_row(D, k - 7, d0);      // ... after the column's last pixel (k, 7): x = 8 j + k - 7
    K.ie[0] = rel_row(D, k - 8, d0);   // up-right after (x - 1, y0 + 8)
    K.rb = K.ie[0];
    K.rt = rel_row(D, k, d0);          // up-right after (k, 0)
  } else if constexpr (PHASE == 1) {
    // D = 128: half-wave 0 of wave w owns chain w + 1 (A then B), half-wave 1 chain 7 - w in the order
    // B, A (both halves then change piece at slot 7 - w); wave 3's second half the main diagonal,
    // cut into 4 + 4 that continue.  D = 256: wave w owns chain w; the main diagonal is one piece.
    const bool main_diag = D == 128 ? (half == 1 && w == 3) : w == 0;
    const int c = D == 128 ? (half == 0 ? w + 1 : (main_diag ? 0 : 7 - w)) : w;
    const Chain ch = chain_of(1, c);
    const int pa = ch.p[0].r * kT + ch.p[0].u, pb = ch.p[1].r * kT + ch.p[1].u;
    if (main_diag) {
      K.pix[0] = 0; K.pix[1] = 4 * kT + 4;
      K.restart1 = false;
      K.ft = rel_row(D, 0, d0); K.fb = rel_row(D, 0, d0);
      if (D == 128) K.ie[1] = rel_row(D, 0, d0);  // (the piece the inverse chain enters at the bottom)
      else K.ie[0] = rel_row(D, 0, d0);
      K.e0_bottom = D != 128;
      K.rb = rel_row(D, 0, d0); K.rt = rel_row(D, 0, d0);
    } else {
      K.carry_first = D == 128 && half == 1;
      K.e0_bottom = K.carry_first;
      K.pix[0] = K.carry_first ? pb : pa;
      K.pix[1] = K.carry_first ? pa : pb;
      K.ft = rel_row(D, c, d0);        // down after (x = 8 j + c, y0 - 1)
      K.fb = rel_row(D, c - 8, d0);    // ... after B's last pixel (c - 1, 7): x = 8 j + c - 8
      const unsigned side = rel_side(D, side_idx_up(kT - c), d0), bot = rel_row(D, c - 8, d0);
      K.ie[0] = K.carry_first ? bot : side;
      K.ie[1] = K.carry_first ? side : bot;
      K.rb = bot;
      K.rt = rel_row(D, c, d0);
      K.rs[0] = K.carry_first ? side : kNoAccess;  // B leaves to the left at (0, 8 - c)
      K.rs[1] = K.carry_first ? kNoAccess : side;
    }
  } else {
    const int c0 = 7 - row;
    const Chain ch = chain_of(2, c0);
    int ul = 0;  // u of the chain's last pixel (row 7)
    unsigned last_entry = kNoAccess;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int q = 0; q < 3; ++q) {
      if (q < ch.np) {
        K.pix[q] = ch.p[q].r * kT + ch.p[q].u;
        const int ue = ch.p[q].u + 2 * (ch.p[q].n - 1), re = ch.p[q].r + ch.p[q].n - 1;  // its last pixel
        if (q == ch.np - 1) {
          ul = ue;
        } else {
          // the inverse chain enters from (ue + 2, re + 1), pixel (ue - 6, re + 1) of the tile on the right
          K.ie[q] = rel_side(D, side_idx_upleft(ue + 2 - kT, re + 1), d0);
        }
        if (q > 0) K.rs[q] = rel_side(D, side_idx_upleft(ch.p[q].u, ch.p[q].r), d0);  // leaves to the left at its first pixel
      }
    }
    last_entry = rel_row(D, ul - 6, d0);     // up-left after (x + 1, y0 + 8)
    K.ft = rel_row(D, c0 - 1, d0);           // down-right after (x - 1, y0 - 1)
    K.fb = rel_row(D, ul - 7, d0);           // ... after the last pixel (ul, 7)
    if (ch.np == 3) K.ie[2] = last_entry; else K.ie[1] = last_entry;
    K.rb = last_entry;
    K.rt = rel_row(D, c0, d0);               // up-left after (c0, 0)
  }
}
STM_HD void chain_io(int D, int phase, int w, int half, int d0, ChainIO &K) {
  if (phase == 0) chain_io_t<0>(D, w, half, d0, K);
  else if (phase == 1) chain_io_t<1>(D, w, half, d0, K);
  else if (phase == 2) chain_io_t<2>(D, w, half, d0, K);
  else chain_io_t<3>(D, w, half, d0, K);
}
// Slots at which a chain of wave w changes piece (the same for both half-waves at D = 128).
STM_HD int na_of(int D, int phase, int w) {
  if (D == 128) return phase == 1 ? 7 - w : (phase == 2 ? w + 1 : kT);
  return phase == 1 ? (w == 0 ? kT : kT - w) : (phase == 2 ? (w >> 1) + 1 : kT);
}
STM_HD int nb_of(int D, int phase, int w) {
  const int g = D == 128 ? w : w >> 1;
  return phase == 2 && g < 3 ? g + 5 : kT;
}

// ---- a tile's terms of the offsets -----------------------------------------------------------------
struct TilePos {
  int i, j;   // band, tile column
  int x0, y0;
  bool gen;   // not a fast tile: validity per lane
};
STM_HD void make_tile_pos(TilePos &T, const SkewGeom &G, int i, int j) {
  T.i = i; T.j = j;
  T.x0 = kT * j; T.y0 = kT * i;
  T.gen = !tile_fast(G, i, j);
}
// Byte offset (inside the image's workspace block) of a row border state given its relative
// offset; kNoAccess when it does not exist (gen tiles only).  need_row7: a state after the
// band's LAST row (the forward pass's bottom exits), which a partial band does not have.
STM_HD unsigned row_state_off(const SkewGeom &G, const TilePos &T, int dir, int t, unsigned rel, bool need_row7) {
  if (rel >= kNoAccess) return kNoAccess;
  if (!T.gen) return rel + (unsigned)(rowb_in_img(G, dir, t, T.x0 - kT) * 4);
  // (D is 128 or 256: a row of 512 or 1,024 bytes -- shifts, not a division by a run-time value)
  const int sh = G.D == 128 ? 9 : 10;
  const int xr = (int)(rel >> sh) - kT, x = T.x0 + xr;
  bool ok = x >= 0 && x < G.W && t >= 0 && t < G.nTR;
  if (need_row7) ok = ok && kT * t + kT - 1 < G.H;
  return ok ? (unsigned)(rowb_in_img(G, dir, t, x) * 4) + (rel & ((1u << sh) - 1u)) : kNoAccess;
}
STM_HD unsigned side_state_off(const SkewGeom &G, const TilePos &T, int jb, unsigned rel) {
  if (rel >= kNoAccess) return kNoAccess;
  const unsigned r = rel & ~1u;
  if (!T.gen) return r + (unsigned)(side_in_img(G, T.i, jb, 0) * 4);
  const bool ok = jb >= 1 && jb < G.nTC;
  return ok ? r + (unsigned)(side_in_img(G, T.i, jb, 0) * 4) : kNoAccess;
}
// An inverse-chain entry of the forward pass: side block of tile j + 1 or row states of band i + 1.
STM_HD unsigned fwd_inverse_entry_off(const SkewGeom &G, const TilePos &T, int dir, unsigned e) {
  if (e >= kNoAccess) return kNoAccess;
  if (!T.gen) {
    const unsigned base = (e & 1u) ? (unsigned)(side_in_img(G, T.i, T.j + 1, 0) * 4) : (unsigned)(rowb_in_img(G, dir, T.i + 1, T.x0 - kT) * 4);
    return (e & ~1u) + base;
  }
  return (e & 1u) ? side_state_off(G, T, T.j + 1, e) : row_state_off(G, T, dir, T.i + 1, e, false);
}

// How the inverse passes of the line through image pixel (gx, gy) count.
STM_HD tiled::Counts line_counts(int phase, int gx, int gy, int H, int W, int true_paths) {
  tiled::TiledGeom T;
  T.H = H; T.W = W;
  tiled::Seg s;
  s.ex = gx; s.ey = gy; s.n = 1;
  return tiled::seg_counts(phase, s, 0, 0, T, true_paths);
}
// ... for a whole tile (as tiled::h2_count_class): 1 = every inverse pass once, 0 = none at all,
// 2 = some once and some not, 3 = the tile holds a line that counts several times or before its
// forward pass.
STM_HD int count_class(int phase, const SkewGeom &G, int i, int j, int true_paths) {
  if (phase < 2 || true_paths) return 1;
  const int e = G.W - G.H;
  if (phase == 2) {
    const int lo = kT * (j - i) - 2 * (kT - 1), hi = kT * (j - i) + (kT - 1);  // x - y over the tile
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
    const int lo = kT * (i + j), hi = kT * (i + j) + (kT - 1);  // x + y over the tile
    if (lo <= G.H - 1 && G.H - 1 <= hi) return 3;
    if (lo >= G.H && hi <= G.W - 1) return 0;
    return (hi < G.H - 1 || lo > G.W - 1) ? 1 : 2;
  }
  return 1;
}

// ---- the re-walked diagonals ------------------------------------------------------------------------
// The reference re-walks one inverse diagonal per family many times: on a wide
// image (e = W - H > 0) the up-left line x - y = e and the up-right line x + y = H - 1 are added
// 1 + e times, on a tall one (e < -1) the up-left line x - y = e is added -e - 1 times BEFORE its
// forward pass.  Float addition has to repeat them one by one, and inside the forward pass those
// repetitions sat on every band's critical path, one band after the other (the lines cross the
// bands in the direction the bands depend on each other): 135 bands x 2 x ~150 us on a 1080p
// image.  The forward pass therefore DEFERS these pixels: it saves what their final sum is made of
// (the sum before pair 2 and the four values of pairs 2 and 3) and the band's owner finishes them
// when the band is done and published, off the pipeline.
// Which lines are deferred: A (pair 2) / B (pair 3).
STM_HD bool defer_line_a(const SkewGeom &G, int true_paths) { return !true_paths && (G.W - G.H > 0 || G.W - G.H < -1); }
STM_HD bool defer_line_b(const SkewGeom &G, int true_paths) { return !true_paths && G.W - G.H > 0; }
STM_HD bool on_line_a(const SkewGeom &G, int x, int y) { return x - y == G.W - G.H; }
STM_HD bool on_line_b(const SkewGeom &G, int x, int y) { return x + y == G.H - 1; }
// Entry of a deferred pixel in its band's block: 0..7 line A by row of the band, 8..15 line B; -1: not deferred.
STM_HD int defer_entry(const SkewGeom &G, int true_paths, int x, int y) {
  if (defer_line_a(G, true_paths) && on_line_a(G, x, y)) return y & (kT - 1);
  if (defer_line_b(G, true_paths) && on_line_b(G, x, y)) return kT + (y & (kT - 1));
  return -1;
}
STM_HD long long defer_in_img(const SkewGeom &G, int t, int entry, int row) {
  return G.defer_in_img + (((long long)t * kDeferPix + entry) * kDeferRows + row) * (long long)G.D;
}
// Pixel of an entry of band t; false: the entry is empty (outside the image, line not deferred, or
// -- line B -- the pixel lies on line A as well and is entry A's).
STM_HD bool defer_pixel(const SkewGeom &G, int true_paths, int t, int entry, int &x, int &y) {
  y = kT * t + (entry & (kT - 1));
  const bool a = entry < kT;
  if (a ? !defer_line_a(G, true_paths) : !defer_line_b(G, true_paths)) return false;
  x = a ? y + (G.W - G.H) : G.H - 1 - y;
  if (!in_image(G, x, y)) return false;
  return a || !(defer_line_a(G, true_paths) && on_line_a(G, x, y));
}

// ---- what a tile reads and writes outside itself ---------------------------------------------
// Tiles of the band above (forward pass) / below (reverse pass) whose row border states a tile
// needs: j and j + 1 / j - 1 and j.  Progress a band's owner waits for before tile j (counted in
// finished tiles from the band's first one in processing order): enough for tile j's own entries
// and for prefetching those of the next tile.
STM_HD int progress_needed(const SkewGeom &G, int jj /* tile's number in processing order */) {
#ifndef STM_SKEW_LEAD
#define STM_SKEW_LEAD 3
#endif
  return imin_(jj + STM_SKEW_LEAD, G.nTC);
}

}  // namespace skew
}  // namespace stm
