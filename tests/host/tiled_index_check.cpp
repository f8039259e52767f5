// Host-side address-range model of the tiled form of the semiglobal aggregation (no GPU).
//
// Evaluates, through the SAME functions the kernels call (csrc/stm_sgm_tiled_index.h), every
// global-memory access the sweep, walk and tile kernels form -- cost rows, intensities, border
// states, mailboxes, outputs, the clamped prefetches past the end of an image, the dummy reads of
// paths that begin at the image border, the running pointers of the interior loops -- for every
// workgroup / wave / row / tile of a grid of shapes, and checks that each lies inside its buffer.
// Also: the workspace regions do not overlap, the tile walks visit every tile exactly once, the
// D = 128 kernel's tile-relative entry offsets agree with the generic addressing, and the running
// pointers of the sweeps agree with the closed forms.
//
//   g++ -O2 -std=c++17 -I stereo-matching_amd/csrc tests/host/tiled_index_check.cpp -o check && ./check
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "stm_sgm_tiled_index.h"

using namespace stm::tiled;

static long long g_checks = 0, g_fail = 0;
static int gH, gW, gD, gB;
#define CHECK(cond, ...)                                                            \
  do {                                                                              \
    ++g_checks;                                                                     \
    if (!(cond)) {                                                                  \
      if (g_fail < 20) {                                                            \
        fprintf(stderr, "FAIL H=%d W=%d D=%d B=%d: %s: ", gH, gW, gD, gB, #cond);   \
        fprintf(stderr, __VA_ARGS__);                                               \
        fprintf(stderr, "\n");                                                      \
      }                                                                             \
      ++g_fail;                                                                     \
    }                                                                               \
  } while (0)

// [lo, lo + len) inside [0, extent)
static void in_range(long long lo, long long len, long long extent, const char *what) {
  CHECK(lo >= 0 && lo + len <= extent, "%s: [%lld, +%lld) not in [0, %lld)", what, lo, len, extent);
}

struct Ext {
  long long cost, pixels, ws;  // floats of one image's cost volume, pixels of an image, workspace floats
};

static void check_layout(const TiledGeom &G, long long total, long long zero) {
  CHECK(G.mbox_off == kHeaderFloats, "mailboxes start behind the header");
  CHECK(G.flag_off == G.mbox_off + 2 * mbox_granules(G), "flags behind the mailboxes");
  CHECK(zero >= G.flag_off + (long long)G.batch * G.nTR * kMaxSegs && zero - G.flag_off - (long long)G.batch * G.nTR * kMaxSegs < 4,
        "zeroed head = header + mailboxes + flags");
  CHECK(kTicketWord < kHeaderFloats, "ticket word inside the header");
  CHECK(G.rowb_off == zero && G.colb_off == G.rowb_off + G.rowb_img * G.batch, "regions in order");
  CHECK(G.colb_off + G.colb_img * G.batch == total, "workspace size");
  CHECK(G.mbox_off % 4 == 0 && G.rowb_off % 4 == 0 && G.colb_off % 4 == 0 && G.rowb_img % 4 == 0 && G.colb_img % 4 == 0,
        "16-byte alignment of the regions");
  CHECK(G.rowb_img >= G.D, "dummy entry read fits");
}

static void check_rowb(const TiledGeom &G, int b, int k, int t, int x, long long off, const char *what) {
  CHECK(k >= 0 && k < 6 && t >= 0 && t < G.nTR && x >= 0 && x < G.W, "%s: indices k=%d t=%d x=%d", what, k, t, x);
  const long long lo = G.rowb_off + (long long)b * G.rowb_img;
  CHECK(off >= lo && off + G.D <= lo + G.rowb_img, "%s: offset %lld outside image %d's row block", what, off, b);
}
static void check_colb(const TiledGeom &G, int b, int k, int tc, int y, long long off, const char *what) {
  CHECK(k >= 0 && k < 6 && tc >= 0 && tc < G.nTC && y >= 0 && y < G.H, "%s: indices k=%d tc=%d y=%d", what, k, tc, y);
  const long long lo = G.colb_off + (long long)b * G.colb_img;
  CHECK(off >= lo && off + G.D <= lo + G.colb_img, "%s: offset %lld outside image %d's column block", what, off, b);
}

static void check_sweeps(const TiledGeom &G, const Ext &E) {
  const int D = G.D, VPL = D / 64, CW = sweep_cw(VPL), NW = kSweepWaves, H = G.H, W = G.W;
  const int PF = 2;
  for (int up = 0; up < 2; ++up)
    for (int b = 0; b < G.batch; ++b)
      for (int s = 0; s < G.strips; ++s)
        for (int wv = 0; wv < NW; ++wv) {
          const int xa = sweep_first_col(s, wv, CW);
          if (xa >= W) continue;  // inactive waves only keep the barriers
          const bool has_left = xa > 0, left_lds = wv > 0, has_right = xa + CW < W, right_lds = wv < NW - 1;
          // mailboxes
          const long long ng = mbox_granules(G);
          if (has_right && !right_lds) {
            in_range(mbox_granule_off(G, up, b, s, 0), (long long)kMboxSlots * D, ng, "pubA");
            CHECK(s + 1 < G.strips, "a right neighbour strip exists");
            in_range(mbox_granule_off(G, up, b, s + 1, 1), (long long)kMboxSlots * D, ng, "subB");
          }
          if (has_left && !left_lds) {
            CHECK(s >= 1, "a left neighbour strip exists");
            in_range(mbox_granule_off(G, up, b, s, 1), (long long)kMboxSlots * D, ng, "pubB");
            in_range(mbox_granule_off(G, up, b, s - 1, 0), (long long)kMboxSlots * D, ng, "subA");
          }
          for (unsigned row = 0; row < (unsigned)H; ++row)
            if (row < 8 || row + 8 > (unsigned)H) in_range(mbox_slot_granule(D, row), D, (long long)kMboxSlots * D, "mailbox slot");
          // D = 128: descriptors over one mailbox, a row's slot as a byte offset, the rows of
          // the whole groups the loop runs (those past H - 1 publish and poll like any other)
          if (VPL == 2) {
            const int run = sweep_rows_run(H, PF);
            CHECK(run >= H && run < H + PF + 1 && (run - 1) % (PF + 1) == 0, "row 0 and whole groups of rows: %d for H=%d", run, H);
            CHECK(mbox_bytes(D) == (unsigned)(kMboxSlots * D) * 8u, "mailbox bytes");
            for (unsigned row = 0; row < (unsigned)run; ++row) {
              if (!(row < 8 || row + 8 > (unsigned)H)) continue;
              CHECK(mbox_slot_byte(D, row) == (unsigned)mbox_slot_granule(D, row) * 8u, "slot byte");
              // lane: granules dh .. dh + 3 (dh = 4 * (lane & 31)), two 16-byte accesses
              for (int hl = 0; hl < 32; ++hl)
                for (int j = 0; j < 4; j += 2)
                  in_range((long long)mbox_slot_byte(D, row) + (unsigned)(hl * 4) * 8u + (unsigned)j * 8u, 16, mbox_bytes(D), "mailbox access (bytes)");
            }
          }
          // generic loop: fetch(t) for t = 0 .. H - 1 + PF + 1
          for (int t = 0; t <= H + PF; ++t) {
            for (int j = 0; j < CW; ++j) in_range(sweep_cost_in_image(G, up, t, xa + j), D, E.cost, "generic cost fetch");
            for (int lane = 0; lane < 64; ++lane) {
              const int pj = imin_(lane / 3, CW - 1), pk = lane % 3, px = imin_(xa + pj, W - 1);
              const int pxs = sweep_pred_col(px, pk, W);
              CHECK(pxs >= 0 && pxs < W, "predecessor column");
              in_range(sweep_intensity_cur(G, up, t, px), 1, E.pixels, "generic intensity (cur)");
              in_range(sweep_intensity_prev(G, up, t, pxs), 1, E.pixels, "generic intensity (prev)");
            }
          }
          // border-state stores, closed forms (generic loop) ...
          for (int t = 0; t < H; ++t) {
            const int y = sweep_row(H, up, t);
            CHECK(y >= 0 && y < H, "row");
            for (int j = 0; j < CW; ++j) {
              const int x = xa + j;
              if (x >= W) continue;
              if (sweep_stores_row(up, y))
                for (int kk = 0; kk < 3; ++kk) {
                  const long long off = rowb_off(G, b, (up ? 3 : 0), y >> 3, x) + kk * rowb_dir_stride(G);
                  CHECK(off == rowb_off(G, b, (up ? 3 : 0) + kk, y >> 3, x), "direction stride");
                  check_rowb(G, b, (up ? 3 : 0) + kk, y >> 3, x, off, "row border store");
                }
              if ((x & (kT - 1)) == kT - 1) check_colb(G, b, up ? 2 : 1, x >> 3, y, colb_off(G, b, up ? 2 : 1, x >> 3, y), "column border store A");
              if ((x & (kT - 1)) == 0) check_colb(G, b, up ? 5 : 4, x >> 3, y, colb_off(G, b, up ? 5 : 4, x >> 3, y), "column border store B");
            }
          }
          // ... and the running pointers of the interior loops (D <= 128: interior waves; D = 128:
          // every active wave, two columns per half-wave)
          const bool interior = VPL <= 2 && xa >= 1 && xa + CW <= W - 1;
          const bool h2 = VPL == 2;
          if (!interior && !h2) continue;
          SweepCursor rows;
          rows.init(G, up, xa);
          for (int n = 0; n <= H + PF; ++n) {  // PF + 1 refills before the loop, one per row
            CHECK(rows.cur_row() == sweep_row(H, up, imin_(n, H - 1)) * W, "cursor row %d", n);
            CHECK(rows.cost_row() == (long long)sweep_row(H, up, imin_(n, H - 1)) * W * D + (long long)xa * D, "cursor cost row %d", n);
            const int want_prev = (n > 0 && H > 1) ? sweep_row(H, up, imin_(n, H - 1) - 1 + (n >= H ? 0 : 0)) : sweep_row(H, up, 0);
            (void)want_prev;
            in_range(rows.prev_row(G), W, E.pixels, "cursor predecessor row");
            in_range(rows.cur_row(), W, E.pixels, "cursor row pixels");
            if (h2) {
              for (int half = 0; half < 2; ++half) {
                const int col0 = xa + 2 * half;
                for (int c = 0; c < 2; ++c) {
                  const long long l = (long long)(imin_(col0 + c, W - 1) - xa) * D;
                  in_range(rows.cost_row() + l, D, E.cost, "h2 cost fetch");
                }
                for (int i16 = 0; i16 < 16; ++i16) {
                  const int pxh = imin_(col0 + imin_(i16 / 3, 1), W - 1), pxsh = sweep_pred_col(pxh, i16 % 3, W);
                  in_range(rows.cur_row() + pxh, 1, E.pixels, "h2 intensity (cur)");
                  in_range(rows.prev_row(G) + pxsh, 1, E.pixels, "h2 intensity (prev)");
                }
              }
            } else {
              in_range(rows.cost_row(), (long long)CW * D, E.cost, "interior cost fetch");
              for (int lane = 0; lane < 64; ++lane) {
                const int pj = imin_(lane / 3, CW - 1), px = imin_(xa + pj, W - 1);
                in_range(rows.cur_row() + px, 1, E.pixels, "interior intensity (cur)");
                in_range(rows.prev_row(G) + sweep_pred_col(px, lane % 3, W), 1, E.pixels, "interior intensity (prev)");
              }
            }
            rows.advance(G);
          }
          // stores of the interior loops: rowV + (y >> 3) * line stride + k * direction stride + column
          const long long rowV = rowb_off(G, b, up ? 3 : 0, 0, xa);
          int ja_col = -1, jb_col = -1;
          for (int j = 0; j < CW; ++j) {
            if (((xa + j) & (kT - 1)) == kT - 1) ja_col = j;
            if (((xa + j) & (kT - 1)) == 0) jb_col = j;
          }
          const int y_first = up ? H - 1 : 0;
          long long colA = -1, colB = -1;
          if (h2) {
            if ((xa & (kT - 1)) == 4 && xa + 3 < W) colA = colb_off(G, b, up ? 2 : 1, (xa + 3) >> 3, y_first);
            if ((xa & (kT - 1)) == 0) colB = colb_off(G, b, up ? 5 : 4, xa >> 3, y_first);
          } else {
            if (ja_col >= 0) colA = colb_off(G, b, up ? 2 : 1, (xa + ja_col) >> 3, y_first);
            if (jb_col >= 0) colB = colb_off(G, b, up ? 5 : 4, (xa + jb_col) >> 3, y_first);
          }
          const int col_step = up ? -colb_row_stride(G) : colb_row_stride(G);
          int y = y_first;
          for (int t = 0; t < H; ++t) {
            if (sweep_stores_row(up, y)) {
              const long long r0 = rowV + (long long)(y >> 3) * rowb_line_stride(G);
              for (int j = 0; j < CW; ++j) {
                if (xa + j >= W) continue;  // (h2 edge waves: predicated; interior waves: all valid)
                for (int kk = 0; kk < 3; ++kk)
                  check_rowb(G, b, (up ? 3 : 0) + kk, y >> 3, xa + j, r0 + kk * rowb_dir_stride(G) + (long long)j * D, "interior row border store");
              }
            }
            if (h2) {  // the column lines as buffer descriptors: base = the line's row 0, the row as a byte offset
              for (int side = 0; side < 2; ++side) {
                const long long at = side == 0 ? colA : colB;
                if (at < 0) continue;
                const long long line0 = (side == 0 ? colb_off(G, b, up ? 2 : 1, (xa + 3) >> 3, y_first) : colb_off(G, b, up ? 5 : 4, xa >> 3, y_first))
                                        - (long long)y_first * colb_row_stride(G);
                CHECK(line0 == (side == 0 ? colb_off(G, b, up ? 2 : 1, (xa + 3) >> 3, 0) : colb_off(G, b, up ? 5 : 4, xa >> 3, 0)), "line base is its row 0");
                CHECK(colb_line_row_byte(G, y) + (unsigned)D * 4u <= colb_line_bytes(G), "row %d inside the line's descriptor", y);
                CHECK(line0 + colb_line_row_byte(G, y) / 4 == at, "descriptor address of row %d equals the running pointer", y);
              }
            }
            if (colA >= 0) {
              const int x = h2 ? xa + 3 : xa + ja_col;
              CHECK(colA == colb_off(G, b, up ? 2 : 1, x >> 3, y), "running column pointer A at row %d", y);
              check_colb(G, b, up ? 2 : 1, x >> 3, y, colA, "interior column border store A");
              colA += col_step;
            }
            if (colB >= 0) {
              const int x = h2 ? xa : xa + jb_col;
              CHECK(colB == colb_off(G, b, up ? 5 : 4, x >> 3, y), "running column pointer B at row %d", y);
              check_colb(G, b, up ? 5 : 4, x >> 3, y, colB, "interior column border store B");
              colB += col_step;
            }
            y += up ? -1 : 1;
          }
        }
}

static void check_hwalk(const TiledGeom &G, const Ext &E) {
  const int PF = 8, W = G.W;
  for (int leftward = 0; leftward < 2; ++leftward)
    for (int i = 0; i < W + PF + 128; ++i) {
      const int x = hwalk_col(W, leftward, i);
      CHECK(x >= 0 && x < W, "walk column");
      in_range((long long)x * G.D, G.D, (long long)W * G.D, "walk cost fetch (within the row)");
    }
  for (int b = 0; b < G.batch; ++b)
    for (int y = 0; y < G.H; ++y)
      for (int x = 0; x < W; ++x) {
        if ((x & (kT - 1)) == kT - 1) check_colb(G, b, 0, x >> 3, y, colb_off(G, b, 0, x >> 3, y), "walk store ->");
        if ((x & (kT - 1)) == 0) check_colb(G, b, 3, x >> 3, y, colb_off(G, b, 3, x >> 3, y), "walk store <-");
      }
  // D <= 128: the states of one image row leave through a descriptor over the row's slots of
  // the direction's column lines (base: tile column 0, one line per tile column further).
  if (G.D <= 128)
    for (int k = 0; k < 6; k += 3)
      for (int y = 0; y < G.H; y += (G.H > 3 ? G.H / 3 : 1)) {
        const long long base = colb_off(G, 0, k, 0, y);
        CHECK(colb_walk_bytes(G) == colb_walk_tc_byte(G, G.nTC - 1) + (unsigned)G.D * 4u, "walk descriptor size");
        for (int tc = 0; tc < G.nTC; ++tc) {
          CHECK(base + colb_walk_tc_byte(G, tc) / 4 == colb_off(G, 0, k, tc, y), "walk descriptor: tile column %d", tc);
          CHECK(colb_walk_tc_byte(G, tc) + (unsigned)G.D * 4u <= colb_walk_bytes(G), "walk descriptor range");
        }
        CHECK((long long)colb_walk_bytes(G) < (1ll << 31), "below the out-of-range marker");
      }
  (void)E;
}

// Border walks: every path pixel fetch inside the image, every store inside its block, every
// pixel on exactly one path per direction, and every border line the tiles read is written.
static void check_bwalk(const TiledGeom &G, const Ext &E) {
  const int PF = 8, H = G.H, W = G.W;
  for (int skip = 0; skip < 2; ++skip) {
    const int n = bwalk_lines(G, skip != 0);
    std::vector<int> visits((size_t)8 * H * W, 0);
    for (int l = 0; l < n; ++l) {
      const WalkLine k = bwalk_line(G, skip != 0, l);
      CHECK(k.dir >= (skip ? 1 : 0) && k.dir < 8 && k.len >= 1, "line %d: dir %d len %d", l, k.dir, k.len);
      CHECK(k.index >= 0 && k.index < bwalk_lines_of_dir(G, k.dir), "line index");
      if (k.dir == 4 || k.dir == 5) CHECK(k.x0 - k.y0 == k.index - (H - 1), "diagonal number = x - y + H - 1");
      if (k.dir == 6 || k.dir == 7) CHECK(k.x0 + k.y0 == k.index, "anti-diagonal number = x + y");
      CHECK(bwalk_line_used(G, k.dir, k.index, 1) && (k.dir == 5 || k.dir == 7 || bwalk_line_used(G, k.dir, k.index, 0)),
            "only inverse diagonals are ever left out");
      CHECK(k.dx == (k.dir == 0 || k.dir == 4 || k.dir == 7 ? 1 : (k.dir == 2 || k.dir == 3 ? 0 : -1)) &&
                k.dy == (k.dir < 2 ? 0 : (k.dir == 2 || k.dir == 4 || k.dir == 6 ? 1 : -1)), "step of direction %d", k.dir);
      const WalkStores st = bwalk_stores(k.dir);
      int x = k.x0, y = k.y0;
      for (int i = 0; i < k.len + PF + 64; ++i) in_range(bwalk_pixel(G, k, i), 1, E.pixels, "walk pixel (incl. prefetch)");
      for (int i = 0; i < k.len; ++i) {
        CHECK(x >= 0 && x < W && y >= 0 && y < H, "walk position");
        CHECK(bwalk_pixel(G, k, i) == (unsigned)(y * W + x), "walk pixel index");
        ++visits[((size_t)k.dir * H + y) * W + x];
        for (int b = 0; b < G.batch; ++b) {
          if (st.rk >= 0 && (y & (kT - 1)) == st.rv) check_rowb(G, b, st.rk, y >> 3, x, rowb_off(G, b, st.rk, y >> 3, x), "walk row store");
          if (st.ck >= 0 && (x & (kT - 1)) == st.cv) check_colb(G, b, st.ck, x >> 3, y, colb_off(G, b, st.ck, x >> 3, y), "walk column store");
        }
        // what the tiles read for this direction at this pixel is what the walk stores there
        bool is_row;
        const bool on_line = (st.rk >= 0 && (y & 7) == st.rv) || (st.ck >= 0 && (x & 7) == st.cv);
        if (on_line) {
          const unsigned off = border_in_image(G, k.dir, x, y, is_row);
          const long long want = is_row ? (long long)rowb_in_image(G, st.rk, y >> 3, x) : (long long)colb_in_image(G, st.ck, x >> 3, y);
          CHECK(is_row ? (st.rk >= 0 && (y & 7) == st.rv) : (st.ck >= 0 && (x & 7) == st.cv), "tiles read a line the walk stores");
          CHECK((long long)off == want, "tiles read where the walk stores (dir %d)", k.dir);
        }
        x += k.dx;
        y += k.dy;
      }
      // the step before the first pixel lies outside the image: the path starts at the border
      CHECK(k.x0 - k.dx < 0 || k.x0 - k.dx >= W || k.y0 - k.dy < 0 || k.y0 - k.dy >= H, "path starts at the image border");
    }
    for (int dir = skip ? 1 : 0; dir < 8; ++dir)
      for (int p = 0; p < H * W; ++p) CHECK(visits[(size_t)dir * H * W + p] == 1, "direction %d visits pixel %d once", dir, p);
  }
}

// Which border lines are ever written (by the sweeps, the walks, the D = 128 tile kernel's exits):
// a tile's entry must point at one of them.
static bool line_written(int dir, int gx, int gy, bool is_row) {
  switch (dir) {
    case 0: return !is_row && (gx & 7) == 7;
    case 1: return !is_row && (gx & 7) == 0;
    case 2: return is_row && (gy & 7) == 7;
    case 3: return is_row && (gy & 7) == 0;
    case 4: return is_row ? (gy & 7) == 7 : (gx & 7) == 7;
    case 5: return is_row ? (gy & 7) == 0 : (gx & 7) == 0;
    case 6: return is_row ? (gy & 7) == 7 : (gx & 7) == 0;
    default: return is_row ? (gy & 7) == 0 : (gx & 7) == 7;
  }
}

static void check_tiles(const TiledGeom &G, const Ext &E) {
  const int D = G.D, H = G.H, W = G.W;
  for (int b = 0; b < G.batch; ++b)
    for (int trow = 0; trow < G.nTR; ++trow)
      for (int tcol = 0; tcol < G.nTC; ++tcol) {
        const int x0 = tcol * kT, y0 = trow * kT;
        const int tw = imin_(kT, W - x0), th = imin_(kT, H - y0);
        CHECK(tw >= 1 && th >= 1, "tile size");
        for (int r = 0; r < kT; ++r)
          for (int c = 0; c < kT; ++c) in_range((long long)tile_cost_pixel(G, x0, y0, tw, th, r, c) * D, D, E.cost, "tile cost prefetch");
        for (int e = 0; e < (kT + 2) * (kT + 2); ++e) in_range(tile_intensity_pixel(G, x0, y0, e), 1, E.pixels, "tile intensity patch");
        for (int pix = 0; pix < kT * kT; ++pix)
          if ((pix & 7) < tw && (pix >> 3) < th) in_range(tile_out_pixel(G, x0, y0, pix), 1, E.pixels, "tile output pixel");
        // entry states, generic forms (one line per wave; D = 128 border tiles: lines 2 w + i)
        int covered[kT * kT] = {0};
        for (int phase = 0; phase < 4; ++phase)
          for (int w = 0; w < kT; ++w) {
            const PhaseGeom g = tile_geom(phase, w, tw, th);
            const Seg segs[2] = {g.a, g.b};
            for (int i = 0; i < (phase >= 2 ? 2 : 1); ++i)
              for (int inv = 0; inv < 2; ++inv) {
                const EntryRef e = tile_entry(G, phase, segs[i], x0, y0, inv != 0);
                in_range(e.off, D, e.is_row ? G.rowb_img : G.colb_img, "tile entry state");
                if (e.inside) {
                  const int dx = phase_dx(phase), dy = phase_dy(phase);
                  const int px = inv ? x0 + segs[i].ex + segs[i].n * dx : x0 + segs[i].ex - dx;
                  const int py = inv ? y0 + segs[i].ey + segs[i].n * dy : y0 + segs[i].ey - dy;
                  CHECK(line_written(2 * phase + inv, px, py, e.is_row), "entry of phase %d at (%d, %d) lies on a stored line", phase, px, py);
                  // an inverse chain that counts (seg_counts) enters from a line the border walks run
                  if (inv && phase >= 2)
                    for (int tp = 0; tp < 2; ++tp) {
                      const Counts c = seg_counts(phase, segs[i], x0, y0, G, tp);
                      const int line = phase == 2 ? (px - py) + (H - 1) : px + py;
                      if (c.inv != 0 || c.pre != 0)
                        CHECK(bwalk_line_used(G, 2 * phase + 1, line, tp), "inverse entry of a counted segment is walked (phase %d, line %d)", phase, line);
                    }
                } else {
                  CHECK(e.off == 0 && e.is_row, "dummy entry = first float of the image's row block");
                }
              }
            if (phase == 0)
              for (int i = 0; i < 2; ++i)
                for (int k = 0; k < segs[i].n; ++k) ++covered[(segs[i].ey + k * phase_dy(phase)) * kT + segs[i].ex + k * phase_dx(phase)];
          }
        for (int pix = 0; pix < kT * kT; ++pix) CHECK(covered[pix] == (((pix & 7) < tw && (pix >> 3) < th) ? 1 : 0), "rows cover the tile once");
        // D = 128: exits of direction 0 (every tile stores where its rows leave it to the right)
        if (D == 128 && tw == kT)
          for (int row = 0; row < th; ++row)
            check_colb(G, b, 0, tcol, y0 + row, colb_off(G, b, 0, tcol, y0) + (long long)row * D, "tile exit store");
        // D = 128, interior tiles: tile-relative entry offsets + the tile's term = the generic address
        if (D == 128 && tile_interior(G, trow, tcol)) {
          CHECK(tw == kT && th == kT, "interior tiles are full");
          for (int phase = 0; phase < 4; ++phase)
            for (int w = 0; w < 4; ++w)
              for (int half = 0; half < 2; ++half) {
                H2Part p[2];
                bool restart;
                h2_parts(phase, w, half, p[0], p[1], restart);
                CHECK(p[0].n + p[1].n == kT && p[0].n == h2_na(phase, w), "a lane's line has eight slots");
                const int dx = phase_dx(phase), dy = phase_dy(phase);
                for (int i = 0; i < 2; ++i) {
                  if (p[i].n == 0) continue;
                  for (int inv = 0; inv < 2; ++inv) {
                    const int rx = inv ? p[i].ex + p[i].n * dx : p[i].ex - dx, ry = inv ? p[i].ey + p[i].n * dy : p[i].ey - dy;
                    bool is_row, is_row2;
                    const long long rel = h2_entry_offset(G, 2 * phase + inv, rx, ry, is_row);
                    const long long abs_ = rel + (is_row ? h2_tile_row_term(G, trow, tcol) : h2_tile_col_term(G, trow, tcol));
                    // (the second part of the main diagonal continues the first: its "entries" are never loaded)
                    const bool continues = phase >= 2 && !restart && ((i == 1 && inv == 0) || (i == 0 && inv == 1));
                    if (continues) continue;
                    const unsigned want = border_in_image(G, 2 * phase + inv, x0 + rx, y0 + ry, is_row2);
                    CHECK(is_row == is_row2 && abs_ == (long long)want, "h2 entry offset, phase %d wave %d half %d", phase, w, half);
                    in_range(abs_, D, is_row ? G.rowb_img : G.colb_img, "h2 entry state");
                    CHECK(line_written(2 * phase + inv, x0 + rx, y0 + ry, is_row), "h2 entry lies on a stored line");
                  }
                }
              }
        }
      }
  // the persistent workgroups' tile walks visit every tile exactly once
  const int ntiles = G.batch * G.nTR * G.nTC;
  for (int stride : {1, 3, G.nTC, G.nTC + 1, 256, 512, ntiles}) {
    if (stride <= 0) continue;
    std::vector<int> seen(ntiles, 0);
    for (int first = 0; first < stride && first < ntiles; ++first) {
      TileWalk t;
      t.init(first, stride, G);
      for (int id = first; id < ntiles; id += stride) {
        CHECK(t.b >= 0 && t.b < G.batch && t.trow >= 0 && t.trow < G.nTR && t.tcol >= 0 && t.tcol < G.nTC, "tile walk position");
        CHECK((t.b * G.nTR + t.trow) * G.nTC + t.tcol == id, "tile walk position = id %d", id);
        ++seen[(t.b * G.nTR + t.trow) * G.nTC + t.tcol];
        if (id + stride < ntiles) t.next(G);
      }
    }
    for (int i = 0; i < ntiles; ++i) CHECK(seen[i] == 1, "tile %d visited once (stride %d)", i, stride);
  }
  for (int by_rows = 0; by_rows < 2; ++by_rows)
    for (int stride : {1, 5, 512}) {
      const int nunits = TileUnitWalk::units(G, by_rows != 0);
      std::vector<int> seen(ntiles, 0);
      for (int first = 0; first < stride && first < nunits; ++first) {
        TileUnitWalk t;
        t.init(first, G, by_rows != 0);
        while (t.unit < nunits) {
          CHECK(t.b >= 0 && t.b < G.batch && t.trow >= 0 && t.trow < G.nTR && t.tcol >= 0 && t.tcol < G.nTC, "unit walk position");
          ++seen[(t.b * G.nTR + t.trow) * G.nTC + t.tcol];
          t.next(G, stride, by_rows != 0);
        }
      }
      for (int i = 0; i < ntiles; ++i) CHECK(seen[i] == 1, "tile %d visited once by the unit walk (rows %d, stride %d)", i, by_rows, stride);
    }
  // row segments: every tile in exactly one unit, units in the order "first segments first",
  // flags inside their region, a segment's left neighbour has a lower unit number
  for (int segs = 1; segs <= kMaxSegs + 1; ++segs) {
    const RowSegs rs = row_segs(G, segs);
    CHECK(rs.nseg >= 1 && rs.nseg <= kMaxSegs && rs.len >= 1, "segments per row");
    std::vector<int> seen(ntiles, 0);
    const int nu = seg_units(G, rs);
    for (int u = 0; u < nu; ++u) {
      const SegUnit q = seg_unit(G, rs, u);
      CHECK(q.b >= 0 && q.b < G.batch && q.trow >= 0 && q.trow < G.nTR && q.seg >= 0 && q.seg < rs.nseg, "unit %d", u);
      CHECK(q.tbeg < q.tend && q.tend <= G.nTC, "unit %d is not empty: [%d, %d)", u, q.tbeg, q.tend);
      CHECK(seg_flag_index(G, q.b, q.trow, q.seg) < (long long)G.batch * G.nTR * kMaxSegs, "flag index");
      if (q.seg > 0) CHECK(u - G.batch * G.nTR >= 0 && seg_unit(G, rs, u - G.batch * G.nTR).tend == q.tbeg, "left neighbour = unit - rows");
      for (int t = q.tbeg; t < q.tend; ++t) ++seen[(q.b * G.nTR + q.trow) * G.nTC + t];
      // where a segment that is not the last of its row leaves its states: column tcol = tend - 1, full width
      if (q.tend < G.nTC) CHECK(q.tend * kT <= G.W, "a segment ends at a full tile");
    }
    for (int i = 0; i < ntiles; ++i) CHECK(seen[i] == 1, "tile %d in one segment (segs %d)", i, segs);
  }
  const int nrows = G.batch * G.nTR;
  for (int stride : {1, 2, 7, 512}) {
    std::vector<int> seen(ntiles, 0);
    for (int first = 0; first < stride && first < nrows; ++first) {
      TileRowWalk t;
      t.init(first, G);
      while (t.row < nrows) {
        ++seen[(t.b * G.nTR + t.trow) * G.nTC + t.tcol];
        t.next(G, stride);
      }
    }
    for (int i = 0; i < ntiles; ++i) CHECK(seen[i] == 1, "tile %d visited once by the row walk (stride %d)", i, stride);
  }
}

int main(int argc, char **argv) {
  const int sides[] = {1, 2, 3, 7, 8, 9, 15, 16, 17, 40, 95, 151, 375, 450};
  const int disps[] = {64, 128, 256};
  const bool quick = argc > 1;  // any argument: a reduced grid
  long long shapes = 0;
  for (int H : sides)
    for (int W : sides)
      for (int D : disps)
        for (int B = 1; B <= 2; ++B) {
          if (quick && (H > 40 || W > 40)) continue;
          gH = H; gW = W; gD = D; gB = B;
          TiledGeom G;
          long long total = 0, zero = 0;
          const bool ok = make_geom(H, W, D, B, G, total, zero);
          CHECK(ok, "make_geom");
          check_layout(G, total, zero);
          Ext E;
          E.cost = (long long)H * W * D;
          E.pixels = (long long)H * W;
          E.ws = total;
          // the sweeps run when the strips fit the device; the index model does not depend on that
          check_sweeps(G, E);
          check_hwalk(G, E);
          check_bwalk(G, E);
          check_tiles(G, E);
          ++shapes;
        }
  printf("%lld shapes, %lld checks, %lld failures\n", shapes, g_checks, g_fail);
  return g_fail ? 1 : 0;
}
