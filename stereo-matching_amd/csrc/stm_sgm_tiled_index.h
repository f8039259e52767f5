// Index model of the tiled form of the semiglobal aggregation (stm_sgm_sweep.hip,
// stm_sgm_hwalk.hip, stm_sgm_tile*.hip): the geometry of its workspace and EVERY expression by
// which its kernels form the address of a global-memory access -- cost rows, intensities,
// border states, mailboxes, outputs, including the clamped prefetches past the end of an image
// and the dummy reads of paths that begin at the image border.
//
// Plain C++, host and device: the kernels call these functions, and tests/host/
// tiled_index_check.cpp (run by tests/test_tiled_index.py, no GPU) evaluates the same functions
// for every workgroup / wave / row / tile of a grid of shapes and asserts that each access lies
// inside its buffer.  Two memory faults of round 2 were out-of-range READS computed from
// interior offsets that only faulted at a mapping boundary; this closes the class instead of
// sampling it.
//
// All offsets are in floats (4-byte elements) unless a name says otherwise.
#pragma once

#if defined(__HIPCC__)
#define STM_HD __host__ __device__ __forceinline__
#else
#define STM_HD inline
#endif

namespace stm {
namespace tiled {

constexpr int kT = 8;            // tile side = spacing of the stored border lines
constexpr int kSweepWaves = 16;  // waves per sweep workgroup (one workgroup per CU)
constexpr int kMboxSlots = 4;    // rows of history per mailbox
constexpr int kHeaderFloats = 64;  // 256 bytes: status word, ticket counter, padding (16-byte alignment of what follows)
constexpr int kTicketWord = 1;     // 32-bit word of the header: the tile kernel's unit counter (row segments)
constexpr int kMaxSegs = 4;        // segments a tile row may be cut into (row_segs)

STM_HD int imin_(int a, int b) { return a < b ? a : b; }
STM_HD int imax_(int a, int b) { return a > b ? a : b; }

// Geometry shared by the kernels.  Offsets in floats from the workspace base.
struct TiledGeom {
  int H, W, D;
  int nTR, nTC;          // tile rows / columns
  long long rowb_off;    // [batch][6][nTR][W][D]: down V,A,B (rows y%8==7), up V,A,B (rows y%8==0)
  long long colb_off;    // [batch][6][nTC][H][D]: ->, dr, ur (columns x%8==7), <-, dl, ul (x%8==0)
  long long rowb_img, colb_img;
  long long mbox_off;    // [2 sweeps][batch][strips][2 sides][kMboxSlots][D] granules (8 bytes each)
  long long flag_off;    // [batch][nTR][kMaxSegs] 32-bit flags: segment of a tile row finished (row_segs)
  int strips;            // column strips per image in the sweeps
  int batch;
};

// Columns a sweep wave owns (24 state registers: 3 directions x CW x VPL) and a strip's width.
STM_HD int sweep_cw(int VPL) { return 8 / VPL; }
STM_HD int sweep_strip_cols(int VPL) { return kSweepWaves * sweep_cw(VPL); }

// Lays the workspace out; returns its size in floats.  `zero_floats`: the head (status word +
// mailboxes) that every call clears.  False when a per-image block does not fit the 32-bit
// offsets the kernels use inside an image.
STM_HD bool make_geom(int H, int W, int D, int batch, TiledGeom &G, long long &total_floats,
                      long long &zero_floats) {
  const int VPL = D / 64;
  G.H = H; G.W = W; G.D = D;
  G.nTR = (H + kT - 1) / kT;
  G.nTC = (W + kT - 1) / kT;
  G.strips = (W + sweep_strip_cols(VPL) - 1) / sweep_strip_cols(VPL);
  G.batch = batch;
  long long off = kHeaderFloats;
  G.mbox_off = off;
  off += 2ll * batch * G.strips * 2 * kMboxSlots * D * 2;  // granules are two floats wide
  G.flag_off = off;
  off += ((long long)batch * G.nTR * kMaxSegs + 3) / 4 * 4;  // (keeps what follows 16-byte aligned)
  zero_floats = off;
  G.rowb_off = off;
  G.rowb_img = 6ll * G.nTR * W * D;
  off += G.rowb_img * batch;
  G.colb_off = off;
  G.colb_img = 6ll * G.nTC * H * D;
  off += G.colb_img * batch;
  total_floats = off;
  // Buffer descriptors and 32-bit offsets address one image's cost volume / border block.
  return (long long)H * W * D * 4 < (1ll << 32) && G.rowb_img * 4 < (1ll << 32) && G.colb_img * 4 < (1ll << 32);
}

// ---- border states --------------------------------------------------------------------------
// Row lines: k = 0..2 down V, A (down-right), B (down-left) at rows y % 8 == 7 (line t = y >> 3);
//            k = 3..5 up V, A (up-right), B (up-left) at rows y % 8 == 0.
// Column lines: k = 0 ->, 1 down-right, 2 up-right at columns x % 8 == 7 (line tc = x >> 3);
//               k = 3 <-, 4 down-left, 5 up-left at columns x % 8 == 0.
STM_HD long long rowb_off(const TiledGeom &G, int b, int k, int t, int x) {
  return G.rowb_off + (long long)b * G.rowb_img + (((long long)k * G.nTR + t) * G.W + x) * (long long)G.D;
}
STM_HD long long colb_off(const TiledGeom &G, int b, int k, int tc, int y) {
  return G.colb_off + (long long)b * G.colb_img + (((long long)k * G.nTC + tc) * G.H + y) * (long long)G.D;
}
// Strides of those layouts: from one row line to the next, from one direction k to the next
// (the sweeps advance running pointers by them), from one image row to the next in a column line.
STM_HD long long rowb_line_stride(const TiledGeom &G) { return (long long)G.W * G.D; }
STM_HD long long rowb_dir_stride(const TiledGeom &G) { return (long long)G.nTR * G.W * G.D; }
STM_HD int colb_row_stride(const TiledGeom &G) { return G.D; }
// One column line (direction k, tile column tc: colb_off(G, b, k, tc, 0) onwards) in bytes, and
// the byte at which image row y begins in it (buffer descriptor over the line).
STM_HD unsigned colb_line_bytes(const TiledGeom &G) { return (unsigned)(G.H * colb_row_stride(G)) * 4u; }
STM_HD unsigned colb_line_row_byte(const TiledGeom &G, int y) { return (unsigned)(y * colb_row_stride(G)) * 4u; }
// The states one image row y keeps in direction k's column lines, one per tile column
// (colb_off(G, b, k, 0, y) onwards, a line apart): the byte at which tile column tc's begins and
// the bytes up to the end of the last one (buffer descriptor of a horizontal walk).
STM_HD unsigned colb_walk_tc_byte(const TiledGeom &G, int tc) { return (unsigned)(tc * G.H * G.D) * 4u; }
STM_HD unsigned colb_walk_bytes(const TiledGeom &G) { return colb_walk_tc_byte(G, G.nTC - 1) + (unsigned)G.D * 4u; }
// The same inside an image's block (32-bit; make_geom has checked that they fit).
STM_HD unsigned rowb_in_image(const TiledGeom &G, int k, int t, int x) {
  return (unsigned)(((k * G.nTR + t) * G.W + x) * G.D);
}
STM_HD unsigned colb_in_image(const TiledGeom &G, int k, int tc, int y) {
  return (unsigned)(((k * G.nTC + tc) * G.H + y) * G.D);
}

// State of direction `dir` (0 right, 1 left, 2 down, 3 up, 4 down-right, 5 up-left, 6
// down-left, 7 up-right) after pixel (gx, gy), which lies on one of the stored lines: offset
// inside the image's row (is_row) or column block.
STM_HD unsigned border_in_image(const TiledGeom &G, int dir, int gx, int gy, bool &is_row) {
  switch (dir) {
    case 0: is_row = false; return colb_in_image(G, 0, gx >> 3, gy);
    case 1: is_row = false; return colb_in_image(G, 3, gx >> 3, gy);
    case 2: is_row = true; return rowb_in_image(G, 0, gy >> 3, gx);
    case 3: is_row = true; return rowb_in_image(G, 3, gy >> 3, gx);
    case 4: is_row = (gy & 7) == 7; return is_row ? rowb_in_image(G, 1, gy >> 3, gx) : colb_in_image(G, 1, gx >> 3, gy);
    case 5: is_row = (gy & 7) == 0; return is_row ? rowb_in_image(G, 5, gy >> 3, gx) : colb_in_image(G, 5, gx >> 3, gy);
    case 6: is_row = (gy & 7) == 7; return is_row ? rowb_in_image(G, 2, gy >> 3, gx) : colb_in_image(G, 4, gx >> 3, gy);
    default: is_row = (gy & 7) == 0; return is_row ? rowb_in_image(G, 4, gy >> 3, gx) : colb_in_image(G, 2, gx >> 3, gy);
  }
}

// ---- mailboxes between column strips ----------------------------------------------------------
// Offset in GRANULES (8 bytes) from the mailbox area's base of mailbox (sweep, image, strip,
// side); a mailbox is kMboxSlots rows of D granules.
STM_HD long long mbox_granule_off(const TiledGeom &G, int sweep, int b, int strip, int side) {
  return ((((long long)sweep * G.batch + b) * G.strips + strip) * 2 + side) * (long long)(kMboxSlots * G.D);
}
STM_HD long long mbox_granules(const TiledGeom &G) { return 2ll * G.batch * G.strips * 2 * kMboxSlots * G.D; }
STM_HD int mbox_slot_granule(int D, unsigned row) { return (int)(row % kMboxSlots) * D; }
// The same in bytes, for a buffer descriptor over ONE mailbox (base: its first granule).
STM_HD unsigned mbox_bytes(int D) { return (unsigned)(kMboxSlots * D) * 8u; }
STM_HD unsigned mbox_slot_byte(int D, unsigned row) { return (unsigned)mbox_slot_granule(D, row) * 8u; }

// ---- row sweeps -------------------------------------------------------------------------------
// Image row of sweep step t (the up sweep runs from the last row).
STM_HD int sweep_row(int H, bool up, int t) { return up ? H - 1 - t : t; }
// A wave's first column; it is active when that lies inside the image.
STM_HD int sweep_first_col(int strip, int wave, int CW) { return (strip * kSweepWaves + wave) * CW; }

// Generic loop: cost row of step t (clamped: steps past the end re-read the last row) at column
// x (clamped: a wave at the right border re-reads its last valid column), inside the image.
STM_HD unsigned sweep_cost_row_in_image(const TiledGeom &G, bool up, int t) {
  return (unsigned)(sweep_row(G.H, up, imin_(t, G.H - 1)) * (G.W * G.D));
}
STM_HD unsigned sweep_cost_col_in_row(const TiledGeom &G, int x) { return (unsigned)(imin_(x, G.W - 1) * G.D); }
STM_HD unsigned sweep_cost_in_image(const TiledGeom &G, bool up, int t, int x) {
  return sweep_cost_row_in_image(G, up, t) + sweep_cost_col_in_row(G, x);
}
// Intensity samples of the penalty of step t for a pixel in column px whose predecessor on the
// path lies in column pxs one row back (both clamped by the caller into [0, W)): pixel indices.
STM_HD unsigned sweep_intensity_cur(const TiledGeom &G, bool up, int t, int px) {
  return (unsigned)(sweep_row(G.H, up, imin_(t, G.H - 1)) * G.W + px);
}
STM_HD unsigned sweep_intensity_prev(const TiledGeom &G, bool up, int t, int pxs) {
  const int tt = imin_(t, G.H - 1);
  return (unsigned)(sweep_row(G.H, up, tt > 0 ? tt - 1 : 0) * G.W + pxs);
}
// Column of the path predecessor for lane role k (0 V, 1 A: from the left, 2 B: from the right).
STM_HD int sweep_pred_col(int px, int k, int W) {
  return k == 1 ? (px > 0 ? px - 1 : 0) : (k == 2 ? imin_(px + 1, W - 1) : px);
}

// Interior loops: a running cursor over the rows instead of a product per row.  The caller
// fetches the row the cursor stands on -- cost_row(): floats inside the image of the wave's first
// column in that row; cur_row() / prev_row(): pixel index of column 0 of the row and of the row
// before it on the path (the first row has none: itself) -- and calls advance().  Past the last
// row the cursor stays on it (those fetches are never used).
struct SweepCursor {
  long long cost;
  int cur;
  int t_next;
  long long row_step;
  int istep;
  STM_HD void init(const TiledGeom &G, bool up, int xa) {
    cost = (long long)(up ? G.H - 1 : 0) * G.W * G.D + (long long)xa * G.D;
    cur = (up ? G.H - 1 : 0) * G.W;
    t_next = 0;
    row_step = (long long)(up ? -1 : 1) * G.W * G.D;
    istep = up ? -G.W : G.W;
  }
  STM_HD long long cost_row() const { return cost; }
  STM_HD int cur_row() const { return cur; }
  STM_HD int prev_row(const TiledGeom &G) const { return t_next > 0 && G.H > 1 ? cur - istep : cur; }
  STM_HD void advance(const TiledGeom &G) {
    if (t_next < G.H - 1) {
      cost += row_step;
      cur += istep;
    }
    ++t_next;
  }
};

// Rows a D = 128 sweep steps through: row 0, then whole groups of pf + 1 (its loop has no test
// between the rows of a group; rows >= H step on the last row's costs again and store nothing).
STM_HD int sweep_rows_run(int H, int pf) { return 1 + (H - 1 + pf) / (pf + 1) * (pf + 1); }

// Border-state rows of a sweep: line index and the float offset of column x in it.
STM_HD bool sweep_stores_row(bool up, int y) { return (y & (kT - 1)) == (up ? 0 : kT - 1); }

// ---- horizontal walks ---------------------------------------------------------------------------
// Column of walk step i (right-to-left walks start at the last column); prefetches past the end
// re-read the last step's column.
STM_HD int hwalk_col(int W, bool leftward, int i) {
  const int j = imin_(i, W - 1);
  return leftward ? W - 1 - j : j;
}

// ---- border walks (the sweep-free way to the border states) -----------------------------------
// One wave walks one TRUE path of one of the eight directions (every pixel lies on exactly one
// path per direction; the reference's quirks for the inverse diagonals are the tile kernel's
// business, which simply does not use what the reference never runs) and stores its state where
// the path leaves a tile.  Lines of an image, in this order (long ones first):
//   direction 0 (right) and 1 (left): H rows each; 2 (down), 3 (up): W columns each;
//   4 (down-right), 5 (up-left): the W + H - 1 diagonals x - y = c, c = -(H - 1) .. W - 1;
//   6 (down-left), 7 (up-right): the W + H - 1 anti-diagonals x + y = s, s = 0 .. W + H - 2.
// skip_right (D = 128): direction 0 is left out (that tile kernel carries it along its tile rows).
struct WalkLine {
  int dir, index;  // direction and the line's number within it
  int x0, y0, dx, dy, len;
};
STM_HD int bwalk_lines_of_dir(const TiledGeom &G, int dir) {
  return dir < 2 ? G.H : (dir < 4 ? G.W : G.W + G.H - 1);
}
STM_HD int bwalk_lines(const TiledGeom &G, bool skip_right) {
  return (skip_right ? 1 : 2) * G.H + 2 * G.W + 4 * (G.W + G.H - 1);
}
STM_HD WalkLine bwalk_line(const TiledGeom &G, bool skip_right, int l) {
  WalkLine k;
  int dir = skip_right ? 1 : 0;
  while (l >= bwalk_lines_of_dir(G, dir)) {
    l -= bwalk_lines_of_dir(G, dir);
    ++dir;
  }
  k.dir = dir;
  k.index = l;
  const int W = G.W, H = G.H;
  int sx, sy, n;  // first pixel and length of the line in its forward direction (0, 2, 4, 6)
  int fdx, fdy;
  if (dir < 2) { sx = 0; sy = l; n = W; fdx = 1; fdy = 0; }
  else if (dir < 4) { sx = l; sy = 0; n = H; fdx = 0; fdy = 1; }
  else if (dir < 6) {
    const int c = l - (H - 1);
    sx = c > 0 ? c : 0; sy = c > 0 ? 0 : -c;
    n = imin_(W - sx, H - sy); fdx = 1; fdy = 1;
  } else {
    const int sum = l;
    sx = imin_(sum, W - 1); sy = sum - sx;
    n = imin_(sx + 1, H - sy); fdx = -1; fdy = 1;
  }
  if (dir & 1) {  // the inverse direction starts at the line's far end
    k.x0 = sx + (n - 1) * fdx; k.y0 = sy + (n - 1) * fdy; k.dx = -fdx; k.dy = -fdy;
  } else {
    k.x0 = sx; k.y0 = sy; k.dx = fdx; k.dy = fdy;
  }
  k.len = n;
  return k;
}
// Lines nobody reads: the reference never runs the inverse passes of some diagonals (SURVEY A.3
// Q6; seg_counts gives them inv = 0 and the tile kernels do not compute those chains), so their
// walks are left out -- on a 1920 x 1080 image 840 of the 2999 up-left and of the up-right lines,
// the full-length ones: 44 % of the pixels of either direction.  `c` / `sum`: the line's x - y /
// x + y.  Nothing is skipped with true_paths.
STM_HD bool bwalk_line_used(const TiledGeom &G, int dir, int l, int true_paths) {
  if (true_paths) return true;
  const int e = G.W - G.H;
  if (dir == 5) {
    const int c = l - (G.H - 1);
    if (e > 0) return !(c >= 0 && c < e);
    if (e < -1) return !(c > e && c < 0);
    return true;
  }
  if (dir == 7 && e > 0) return !(l >= G.H && l <= G.W - 1);
  return true;
}
// Pixel index (inside the image) of step i of a line; prefetches past its end re-read the last.
STM_HD unsigned bwalk_pixel(const TiledGeom &G, const WalkLine &k, int i) {
  const int j = imin_(i, k.len - 1);
  return (unsigned)(k.y0 * G.W + k.x0 + j * (k.dy * G.W + k.dx));
}
// Where direction `dir` keeps its state: on row lines (y % 8 == rv, block index rk) and / or
// column lines (x % 8 == cv, block index ck); -1: none.  = what border_in_image reads.
struct WalkStores {
  int rk, rv, ck, cv;
};
STM_HD WalkStores bwalk_stores(int dir) {
  switch (dir) {
    case 0: return {-1, 0, 0, 7};
    case 1: return {-1, 0, 3, 0};
    case 2: return {0, 7, -1, 0};
    case 3: return {3, 0, -1, 0};
    case 4: return {1, 7, 1, 7};
    case 5: return {5, 0, 5, 0};
    case 6: return {2, 7, 4, 0};
    default: return {4, 0, 2, 7};
  }
}

// ---- tiles ----------------------------------------------------------------------------------------
struct Seg {
  int ex, ey, n;  // first pixel in F order (tile coordinates) and pixel count
};
struct PhaseGeom {
  Seg a, b;
};
// Step of the forward direction of pair PHASE (0 right/left, 1 down/up, 2 down-right/up-left,
// 3 down-left/up-right).
STM_HD constexpr int phase_dx(int phase) { return phase == 0 ? 1 : (phase == 1 ? 0 : (phase == 2 ? 1 : -1)); }
STM_HD constexpr int phase_dy(int phase) { return phase == 0 ? 0 : 1; }

// Line k of a pair inside a tw x th tile: row k, column k, or diagonal k of 15.
STM_HD Seg tile_line(int phase, int k, int tw, int th) {
  Seg s;
  if (phase == 0) { s.ex = 0; s.ey = k; s.n = k < th ? tw : 0; }
  else if (phase == 1) { s.ex = k; s.ey = 0; s.n = k < tw ? th : 0; }
  else if (phase == 2) {  // x - y = k - 7, walked (+1, +1)
    const int c = k - (kT - 1);
    const int xlo = c > 0 ? c : 0, xhi = imin_(tw - 1, th - 1 + c);
    s.ex = xlo; s.ey = xlo - c; s.n = xhi >= xlo ? xhi - xlo + 1 : 0;
  } else {                // x + y = k, walked (-1, +1)
    const int xlo = k - th + 1 > 0 ? k - th + 1 : 0, xhi = imin_(tw - 1, k);
    s.ex = xhi; s.ey = k - xhi; s.n = xhi >= xlo ? xhi - xlo + 1 : 0;
  }
  if (s.n <= 0) { s.ex = 0; s.ey = 0; s.n = 0; }
  return s;
}
// The (up to two) segments wave w owns in a pair: line w, and for the diagonal pairs line w + 8.
STM_HD PhaseGeom tile_geom(int phase, int w, int tw, int th) {
  PhaseGeom g;
  g.a = tile_line(phase, w, tw, th);
  if (phase >= 2 && w < kT - 1) g.b = tile_line(phase, w + kT, tw, th);
  else { g.b.ex = 0; g.b.ey = 0; g.b.n = 0; }
  if (g.a.n == 0) { g.a = g.b; g.b.n = 0; }  // keep the slots dense
  return g;
}

// Entry states of a segment: the forward chain enters from the pixel before its first one, the
// inverse chain from the pixel behind its last one.  `inside` false: the path begins at the
// image border (L = C) and nothing real is loaded -- the kernels then read a valid dummy (the
// first float of the image's row border block, offset 0 with is_row = true).
struct EntryRef {
  unsigned off;
  bool is_row;
  bool inside;
};
STM_HD EntryRef tile_entry(const TiledGeom &G, int phase, const Seg &s, int x0, int y0, bool inverse) {
  const int dx = phase_dx(phase), dy = phase_dy(phase);
  const int px = inverse ? x0 + s.ex + s.n * dx : x0 + s.ex - dx;
  const int py = inverse ? y0 + s.ey + s.n * dy : y0 + s.ey - dy;
  EntryRef e;
  e.inside = s.n > 0 && px >= 0 && px < G.W && py >= 0 && py < G.H;
  e.is_row = true;
  e.off = 0;
  if (e.inside) e.off = border_in_image(G, 2 * phase + (inverse ? 1 : 0), px, py, e.is_row);
  return e;
}

// Prefetch of a tile's cost rows: tile row r (clamped into a partial tile), tile column c
// (clamped): pixel index inside the image.
STM_HD unsigned tile_cost_pixel(const TiledGeom &G, int x0, int y0, int tw, int th, int r, int c) {
  return (unsigned)((y0 + imin_(r, th - 1)) * G.W + x0 + imin_(c, tw - 1));
}
// Element e (0 .. 99, clamped by the caller) of the 10 x 10 intensity patch around a tile: pixel
// index inside the image, clamped at the image border.
STM_HD unsigned tile_intensity_pixel(const TiledGeom &G, int x0, int y0, int e) {
  const int r = e / (kT + 2) - 1, c = e % (kT + 2) - 1;
  const int gy = imin_(imax_(y0 + r, 0), G.H - 1), gx = imin_(imax_(x0 + c, 0), G.W - 1);
  return (unsigned)(gy * G.W + gx);
}
// Output pixel of tile pixel pix (y * 8 + x).
STM_HD unsigned tile_out_pixel(const TiledGeom &G, int x0, int y0, int pix) {
  return (unsigned)((y0 + (pix >> 3)) * G.W + (x0 + (pix & (kT - 1))));
}

// Position of a tile for a persistent workgroup that takes tiles first, first + stride, ...
// (row-major over tiles, images one after the other), advanced without divisions.
struct TileWalk {
  int b, trow, tcol;
  int drow, dcol;  // stride = drow * nTC + dcol
  STM_HD void init(int first, int stride, const TiledGeom &G) {
    const int per_image = G.nTR * G.nTC;
    b = first / per_image;
    const int t = first - b * per_image;
    trow = t / G.nTC;
    tcol = t - trow * G.nTC;
    drow = stride / G.nTC;
    dcol = stride - drow * G.nTC;
  }
  STM_HD void next(const TiledGeom &G) {
    tcol += dcol;
    trow += drow;
    if (tcol >= G.nTC) { tcol -= G.nTC; ++trow; }
    while (trow >= G.nTR) { trow -= G.nTR; ++b; }
  }
};
// ... and for a workgroup that owns whole tile rows (rows first, first + stride, ... of
// batch * nTR) and walks each from left to right (D = 128).
struct TileRowWalk {
  int b, trow, tcol, row;
  STM_HD void init(int r, const TiledGeom &G) {
    row = r;
    b = r / G.nTR;
    trow = r - b * G.nTR;
    tcol = 0;
  }
  STM_HD void next(const TiledGeom &G, int stride) {
    if (++tcol == G.nTC) init(row + stride, G);
  }
};
// ... or either, chosen at run time (the D = 128 tile kernel: whole tile rows when it carries the
// left-to-right state along them, single tiles when that state comes from a walk -- a single
// pair has 47 tile rows for 512 workgroups).  `unit` counts rows or tiles.
struct TileUnitWalk {
  int b, trow, tcol, unit;
  STM_HD static int units(const TiledGeom &G, bool by_rows) { return G.batch * G.nTR * (by_rows ? 1 : G.nTC); }
  STM_HD void init(int u, const TiledGeom &G, bool by_rows) {
    unit = u;
    if (by_rows) {
      b = u / G.nTR;
      trow = u - b * G.nTR;
      tcol = 0;
    } else {
      const int per_image = G.nTR * G.nTC;
      b = u / per_image;
      const int t = u - b * per_image;
      trow = t / G.nTC;
      tcol = t - trow * G.nTC;
    }
  }
  STM_HD void next(const TiledGeom &G, int stride, bool by_rows) {
    if (by_rows) {
      if (++tcol == G.nTC) init(unit + stride, G, true);
    } else {
      init(unit + stride, G, false);
    }
  }
};
// Tile rows cut into segments (the D = 64 / 256 tile kernel behind row sweeps): a workgroup takes
// a segment, walks its tiles left to right and carries the left-to-right state in registers,
// so that direction needs no walk of its own -- one read of the cost volume fewer.  A whole row
// per workgroup would leave 1,080 rows of eight 1080p pairs to 256 workgroups (five rounds for
// 4.2 rounds of work); segments are units of a finer grain, handed out by an atomic ticket in
// the order "all first segments, all second segments, ...".  A segment that is not the first of
// its row starts from the state its left neighbour left in the column border states and waits
// for that neighbour's flag; the neighbour's ticket is lower, so it is held by a workgroup that
// is running or done (every unit is taken by ticket, a workgroup's first one too), and first
// segments wait for nobody: no residency requirement, no deadlock.
struct RowSegs {
  int len, nseg;  // tiles per segment, segments per row (the last one may be shorter)
};
STM_HD RowSegs row_segs(const TiledGeom &G, int segs) {
  RowSegs r;
  const int want = imax_(1, imin_(segs, kMaxSegs));
  r.len = (G.nTC + want - 1) / want;
  r.nseg = (G.nTC + r.len - 1) / r.len;
  return r;
}
struct SegUnit {
  int b, trow, seg, tbeg, tend;  // image, tile row, segment and its tile columns [tbeg, tend)
};
STM_HD int seg_units(const TiledGeom &G, const RowSegs &r) { return r.nseg * G.batch * G.nTR; }
STM_HD SegUnit seg_unit(const TiledGeom &G, const RowSegs &r, int unit) {
  const int nrows = G.batch * G.nTR;
  SegUnit u;
  u.seg = unit / nrows;
  const int row = unit - u.seg * nrows;
  u.b = row / G.nTR;
  u.trow = row - u.b * G.nTR;
  u.tbeg = u.seg * r.len;
  u.tend = imin_(G.nTC, u.tbeg + r.len);
  return u;
}
// Index of a segment's flag (32-bit words from G.flag_off).
STM_HD long long seg_flag_index(const TiledGeom &G, int b, int trow, int seg) {
  return ((long long)b * G.nTR + trow) * kMaxSegs + seg;
}

// A tile all of whose neighbours exist (every line enters from a border state).
STM_HD bool tile_interior(const TiledGeom &G, int trow, int tcol) {
  return trow >= 1 && trow <= G.nTR - 2 && tcol >= 1 && tcol <= G.nTC - 2 && (trow + 1) * kT <= G.H &&
         (tcol + 1) * kT <= G.W;
}

// ---- D = 128: two lines of a pair per wave ---------------------------------------------------
// A wave's two lines are chosen so that both change segment at the same slot (line w in the
// order (first, second) with line 6 - w in the order (second, first), w = 0..2; wave 3: line 3
// and the main diagonal, cut into 4 + 4 that continue).
struct H2Part {
  int ex, ey, n;
};
STM_HD void h2_parts(int phase, int w, int half, H2Part &p0, H2Part &p1, bool &restart) {
  const int dx = phase_dx(phase), dy = phase_dy(phase);
  const Seg none = {0, 0, 0};
  Seg a = none, b = none;
  if (phase < 2) {
    a = tile_line(phase, 2 * w + half, kT, kT);
    restart = false;
  } else if (w < 3) {
    const int j = half ? 6 - w : w;
    a = tile_line(phase, half ? j + kT : j, kT, kT);
    b = tile_line(phase, half ? j : j + kT, kT, kT);
    restart = true;
  } else if (!half) {
    a = tile_line(phase, 3, kT, kT);
    b = tile_line(phase, 3 + kT, kT, kT);
    restart = true;
  } else {
    a = tile_line(phase, kT - 1, kT, kT);
    a.n = 4;
    b.ex = a.ex + 4 * dx; b.ey = a.ey + 4 * dy; b.n = 4;
    restart = false;
  }
  p0.ex = a.ex; p0.ey = a.ey; p0.n = a.n;
  p1.ex = b.ex; p1.ey = b.ey; p1.n = b.n;
}
// Slot at which a wave's lines change segment.
STM_HD constexpr int h2_na(int phase, int w) { return phase < 2 ? kT : (w < 3 ? w + 1 : 4); }

// Offset of the state of direction `dir` after the pixel (rx, ry) RELATIVE to a tile's origin,
// rx, ry in [-1, 8], inside an image's block of row (is_row) or column border states, minus the
// tile's own term (h2_tile_row_term / h2_tile_col_term).  Only meaningful for interior tiles:
// added to anything else it may leave the workspace (e.g. by -(H + 1) rows of D for a tall image).
STM_HD int h2_entry_offset(const TiledGeom &G, int dir, int rx, int ry, bool &is_row) {
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
STM_HD long long h2_tile_row_term(const TiledGeom &G, int trow, int tcol) { return (long long)(trow * G.W + tcol * kT) * G.D; }
STM_HD long long h2_tile_col_term(const TiledGeom &G, int trow, int tcol) { return (long long)(tcol * G.H + trow * kT) * G.D; }

// How the inverse passes of a diagonal count (following the reference's stored path ends,
// SURVEY A.3 Q6): `inv` times after the forward pass and `pre` times before it.
struct Counts {
  int inv, pre;
};
STM_HD Counts seg_counts(int phase, const Seg &s, int x0, int y0, const TiledGeom &G, int true_paths) {
  Counts c;
  c.inv = 1;
  c.pre = 0;
  if (phase < 2 || true_paths || s.n == 0) return c;
  const int e = G.W - G.H;
  const int gx = x0 + s.ex, gy = y0 + s.ey;
  if (phase == 2) {
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
// ... for a whole tile: 1 = every inverse pass once, 0 = none at all, 2 = some once and some not
// (per line), 3 = the tile holds the diagonal that counts several times or before its forward pass.
STM_HD int h2_count_class(int phase, const TiledGeom &G, int x0, int y0, int true_paths) {
  if (phase < 2 || true_paths) return 1;
  const int e = G.W - G.H;
  if (phase == 2) {
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

}  // namespace tiled
}  // namespace stm
