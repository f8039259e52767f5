// Host-side DATAFLOW check of the skewed-tile form of the semiglobal aggregation
// (csrc/stm_sgm_skew_index.h: the functions the kernels call to place every state they hand over).
// No GPU.  For a grid of shapes and both disparity counts it replays, symbolically, what the two
// passes store and load:
//   * every store of a state -- row border states, side blocks -- gets the label (direction, pixel
//     after which that direction's state was taken, producing tile); two different labels on one
//     address are an error, an address outside the image's workspace block is an error;
//   * every load of an entry state whose path really comes from inside the image must find, at its
//     address, exactly the label the recurrence needs (the right direction after the right pixel),
//     and its producer must be one the progress rule (progress_needed) has waited for;
//   * loads whose value the kernels do not use (paths that begin at the image border) must still
//     be in range or carry the "no access" offset;
//   * the pieces of a chain tile the 64 pixels of a tile once per pair, and a piece's exit to the
//     right is the next piece's entry one tile further on (what the owner carries in registers).
// Prints "<shapes> shapes, <checks> checks, 0 failures".
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "stm_sgm_skew_index.h"

using namespace stm::skew;

static long long g_checks = 0, g_failures = 0;
#define CHECK(cond, ...)                                      \
  do {                                                        \
    ++g_checks;                                               \
    if (!(cond)) {                                            \
      if (++g_failures <= 20) { std::printf("FAIL " __VA_ARGS__); std::printf("\n"); } \
    }                                                         \
  } while (0)

struct Label {
  int dir, x, y, ti, tj;  // direction (0..7 as in the kernels' pairs: 2 * pair + inverse), pixel, producing tile
};
static bool same(const Label &a, const Label &b) { return a.dir == b.dir && a.x == b.x && a.y == b.y; }

struct Shape {
  int H, W, D;
};

// Image pixel of tile pixel `pix` (r * 8 + u).
static void pixel_of(const TilePos &T, int pix, int &x, int &y) {
  const int u = pix & 7, r = pix >> 3;
  x = T.x0 + u - r;
  y = T.y0 + r;
}

static void check_shape(const Shape &S) {
  SkewGeom G;
  long long total = 0, zero = 0;
  if (!make_geom(S.H, S.W, S.D, 1, G, total, zero)) { CHECK(false, "make_geom %d %d %d", S.H, S.W, S.D); return; }
  const int D = S.D, waves = D == 128 ? 4 : 8, halves = D == 128 ? 2 : 1;
  const unsigned block_bytes = (unsigned)(G.img_floats * 4), row_bytes = (unsigned)D * 4u;
  CHECK(G.img_floats * 4 < (long long)kNoAccess, "image block too large");
  // ---- chains tile a tile; carries continue ------------------------------------------------------
  for (int phase = 0; phase < 4; ++phase) {
    int seen[64] = {0};
    for (int k = 0; k < 8; ++k) {
      const Chain c = chain_of(phase, k);
      int slots = 0;
      for (int q = 0; q < c.np; ++q) {
        for (int t = 0; t < c.p[q].n; ++t) {
          const int u = c.p[q].u + t * tile_du(phase), r = c.p[q].r + t * tile_dr(phase);
          CHECK(u >= 0 && u < 8 && r >= 0 && r < 8, "pair %d chain %d piece %d leaves the tile", phase, k, q);
          if (u >= 0 && u < 8 && r >= 0 && r < 8) ++seen[r * 8 + u];
          ++slots;
        }
        if (q + 1 < c.np) {  // exit to the right = the next piece's entry in the next tile
          const int ue = c.p[q].u + c.p[q].n * tile_du(phase) - 8, re = c.p[q].r + c.p[q].n * tile_dr(phase);
          CHECK(ue == c.p[q + 1].u && re == c.p[q + 1].r, "pair %d chain %d: piece %d does not continue as piece %d", phase, k, q, q + 1);
        }
      }
      CHECK(slots == 8, "pair %d chain %d has %d slots", phase, k, slots);
    }
    for (int p = 0; p < 64; ++p) CHECK(seen[p] == 1, "pair %d: pixel %d covered %d times", phase, p, seen[p]);
  }
  // ---- the lanes' chains: slot counts agree with na_of / nb_of -------------------------------------
  std::map<unsigned, Label> mem;  // byte offset in the image's block -> what was stored there
  auto store = [&](unsigned off, const Label &l, const char *what) {
    if (off >= kNoAccess) return;
    CHECK(off % row_bytes == 0 && off + row_bytes <= block_bytes, "%s: store at %u outside the block (%u) %dx%dx%d tile %d,%d", what, off,
          block_bytes, S.H, S.W, D, l.ti, l.tj);
    auto it = mem.find(off);
    if (it != mem.end()) CHECK(same(it->second, l), "%s: two states on address %u (dir %d pixel %d,%d and dir %d pixel %d,%d)", what, off,
                               it->second.dir, it->second.x, it->second.y, l.dir, l.x, l.y);
    mem[off] = l;
  };
  // ---- reverse pass: stores --------------------------------------------------------------------------
  for (int i = 0; i < G.nTR; ++i)
    for (int j = 0; j < G.nTC; ++j) {
      TilePos T;
      make_tile_pos(T, G, i, j);
      for (int w = 0; w < waves; ++w)
        for (int half = 0; half < halves; ++half)
          for (int phase = 0; phase < 4; ++phase) {
            ChainIO K;
            chain_io(D, phase, w, half, 0, K);
            const int na = na_of(D, phase, w), nb = nb_of(D, phase, w), np = nb < 8 ? 3 : (na < 8 ? 2 : 1);
            const int inv_dir = 2 * phase + 1;
            // top exit: the state after the first pixel of the piece that enters at the tile's top
            if (K.rt < kNoAccess) {
              int top_piece = 0;
              if (phase == 1) top_piece = K.carry_first ? 1 : 0;
              int x, y;
              pixel_of(T, K.pix[top_piece], x, y);
              if (phase == 1 && !K.restart1) pixel_of(T, K.pix[0], x, y);  // the main diagonal
              CHECK((K.pix[phase == 1 && !K.restart1 ? 0 : top_piece] >> 3) == 0, "top exit not in row 0");
              store(row_state_off(G, T, 3 + (phase == 1 ? 0 : (phase == 2 ? 1 : 2)), T.i, K.rt, false), Label{inv_dir, x, y, i, j}, "reverse top");
            }
            for (int q = 0; q < np; ++q)
              if (K.rs[q] < kNoAccess) {
                int x, y;
                pixel_of(T, K.pix[q], x, y);
                store(side_state_off(G, T, T.j, K.rs[q]), Label{inv_dir, x, y, i, j}, "reverse side");
              }
          }
    }
  // ---- forward pass: loads of the inverse chains' entries -------------------------------------------
  for (int i = 0; i < G.nTR; ++i)
    for (int j = 0; j < G.nTC; ++j) {
      TilePos T;
      make_tile_pos(T, G, i, j);
      for (int w = 0; w < waves; ++w)
        for (int half = 0; half < halves; ++half)
          for (int phase = 0; phase < 4; ++phase) {
            ChainIO K;
            chain_io(D, phase, w, half, 0, K);
            const int na = na_of(D, phase, w), nb = nb_of(D, phase, w), np = nb < 8 ? 3 : (na < 8 ? 2 : 1);
            const int start[3] = {0, na, nb}, end[3] = {na, np > 1 ? nb : 8, 8};
            const int dirs_row[4] = {0, 3, 4, 5};
            for (int q = 0; q < np; ++q) {
              if (phase == 1 && !K.restart1 && q == 0) continue;  // the main diagonal's first half continues
              const int n = end[q] - start[q];
              int x, y;
              pixel_of(T, K.pix[q], x, y);
              const int sx = x + n * img_dx(phase), sy = y + n * img_dy(phase);  // the forward successor of the piece's last pixel
              const unsigned off = fwd_inverse_entry_off(G, T, dirs_row[phase], K.ie[q]);
              CHECK(off >= kNoAccess || (off % row_bytes == 0 && off + row_bytes <= block_bytes), "forward inverse entry out of range");
              const bool last_valid = in_image(G, sx - img_dx(phase), sy - img_dy(phase));
              if (in_image(G, sx, sy) && last_valid) {
                auto it = mem.find(off);
                CHECK(it != mem.end(), "%dx%dx%d tile %d,%d pair %d wave %d half %d piece %d: nothing stored at its inverse entry", S.H, S.W, D, i, j,
                      phase, w, half, q);
                if (it != mem.end()) {
                  CHECK(it->second.dir == 2 * phase + 1 && it->second.x == sx && it->second.y == sy,
                        "%dx%dx%d tile %d,%d pair %d piece %d: inverse entry holds dir %d after %d,%d, wanted after %d,%d", S.H, S.W, D, i, j, phase,
                        q, it->second.dir, it->second.x, it->second.y, sx, sy);
                }
              }
            }
          }
    }
  // ---- reverse pass: bottom entries against its own top stores, and the progress rule ------------------
  for (int i = 0; i + 1 < G.nTR; ++i)
    for (int j = 0; j < G.nTC; ++j) {
      TilePos T;
      make_tile_pos(T, G, i, j);
      const int jj = G.nTC - 1 - j;
      for (int w = 0; w < waves; ++w)
        for (int half = 0; half < halves; ++half)
          for (int phase = 1; phase < 4; ++phase) {
            ChainIO K;
            chain_io(D, phase, w, half, 0, K);
            if (K.rb >= kNoAccess) continue;
            const int na = na_of(D, phase, w), nb = nb_of(D, phase, w), np = nb < 8 ? 3 : (na < 8 ? 2 : 1);
            // the piece that leaves at the bottom: the last one in natural order
            int q = np - 1;
            if (phase == 1 && K.carry_first) q = 0;
            const int start[3] = {0, na, nb}, end[3] = {na, np > 1 ? nb : 8, 8};
            int x, y;
            pixel_of(T, K.pix[q], x, y);
            const int n = end[q] - start[q];
            const int lx = x + (n - 1) * img_dx(phase), ly = y + (n - 1) * img_dy(phase);
            CHECK(phase == 1 && !K.restart1 ? true : (ly == T.y0 + 7), "bottom piece does not end in row 7");
            const int sx = (phase == 1 && !K.restart1) ? T.x0 + 1 : lx + img_dx(phase), sy = T.y0 + 8;  // main diagonal: after (7,7) comes (8,8): x0 + 7 - 7 + ... = x0 + 1?
            const int mx = (phase == 1 && !K.restart1) ? T.x0 + 0 + img_dx(phase) * 0 : sx;
            (void)mx;
            const int dir_row = phase == 1 ? 3 : (phase == 2 ? 4 : 5);
            const unsigned off = row_state_off(G, T, dir_row, T.i + 1, K.rb, false);
            // successor pixel for the main diagonal: last pixel (7,7) -> image (x0, y0 + 7); successor (x0, y0 + 8)
            const int ssx = (phase == 1 && !K.restart1) ? T.x0 : sx;
            if (in_image(G, ssx, sy) && in_image(G, ssx - img_dx(phase), sy - img_dy(phase))) {
              auto it = mem.find(off);
              CHECK(it != mem.end(), "reverse bottom entry: nothing stored (tile %d,%d pair %d)", i, j, phase);
              if (it != mem.end()) {
                CHECK(it->second.dir == 2 * phase + 1 && it->second.x == ssx && it->second.y == sy, "reverse bottom entry: wrong state (tile %d,%d pair %d)", i, j,
                      phase);
                // requested one tile ahead (at tile jj - 1): the producer must be among the tiles the rule has waited for
                const int pjj = G.nTC - 1 - it->second.tj;
                CHECK(it->second.ti == i + 1 && pjj < progress_needed(G, jj > 0 ? jj - 1 : 0), "reverse bottom entry of tile %d (order %d) produced by order %d: beyond progress_needed",
                      j, jj, pjj);
              }
            }
          }
    }
  // ---- forward pass: bottom exits and top entries, and the progress rule --------------------------------
  std::map<unsigned, Label> down;
  for (int i = 0; i < G.nTR; ++i)
    for (int j = 0; j < G.nTC; ++j) {
      TilePos T;
      make_tile_pos(T, G, i, j);
      for (int w = 0; w < waves; ++w)
        for (int half = 0; half < halves; ++half)
          for (int phase = 1; phase < 4; ++phase) {
            ChainIO K;
            chain_io(D, phase, w, half, 0, K);
            if (K.fb >= kNoAccess) continue;
            const int na = na_of(D, phase, w), nb = nb_of(D, phase, w), np = nb < 8 ? 3 : (na < 8 ? 2 : 1);
            int q = np - 1;
            if (phase == 1 && K.carry_first) q = 0;
            const int start[3] = {0, na, nb}, end[3] = {na, np > 1 ? nb : 8, 8};
            int x, y;
            pixel_of(T, K.pix[q], x, y);
            int n = end[q] - start[q];
            int lx = x + (n - 1) * img_dx(phase), ly = y + (n - 1) * img_dy(phase);
            if (phase == 1 && !K.restart1) { lx = T.x0; ly = T.y0 + 7; }  // the main diagonal ends at (7, 7)
            const unsigned off = row_state_off(G, T, phase - 1, T.i, K.fb, true);
            if (off >= kNoAccess) continue;
            CHECK(off % row_bytes == 0 && off + row_bytes <= block_bytes, "forward bottom exit out of range");
            CHECK(mem.find(off) == mem.end(), "forward bottom exit on a reverse pass state");
            auto it = down.find(off);
            const Label l{2 * phase, lx, ly, i, j};
            if (it != down.end()) CHECK(same(it->second, l), "two forward states on one address");
            down[off] = l;
          }
    }
  for (int i = 1; i < G.nTR; ++i)
    for (int j = 0; j < G.nTC; ++j) {
      TilePos T;
      make_tile_pos(T, G, i, j);
      for (int w = 0; w < waves; ++w)
        for (int half = 0; half < halves; ++half)
          for (int phase = 1; phase < 4; ++phase) {
            ChainIO K;
            chain_io(D, phase, w, half, 0, K);
            if (K.ft >= kNoAccess) continue;
            // the piece that enters at the top: its first pixel lies in row 0
            int q = 0;
            if (phase == 1 && K.carry_first) q = 1;
            int x, y;
            pixel_of(T, K.pix[q], x, y);
            if (phase == 1 && !K.restart1) { x = T.x0; y = T.y0; }
            CHECK(y == T.y0, "top piece does not begin in row 0 (pair %d wave %d half %d)", phase, w, half);
            const int px = x - img_dx(phase), py = y - img_dy(phase);
            const unsigned off = row_state_off(G, T, phase - 1, T.i - 1, K.ft, false);
            CHECK(off >= kNoAccess || (off % row_bytes == 0 && off + row_bytes <= block_bytes), "forward top entry out of range");
            if (in_image(G, px, py) && in_image(G, x, y)) {
              auto it = down.find(off);
              CHECK(it != down.end(), "%dx%dx%d tile %d,%d pair %d: nothing stored at its top entry", S.H, S.W, D, i, j, phase);
              if (it != down.end()) {
                CHECK(it->second.dir == 2 * phase && it->second.x == px && it->second.y == py, "%dx%dx%d tile %d,%d pair %d: top entry holds the state after %d,%d, wanted %d,%d",
                      S.H, S.W, D, i, j, phase, it->second.x, it->second.y, px, py);
                // pair 1's entry is requested a tile ahead, the others at the tile itself
                const int at = phase == 1 && j > 0 ? j - 1 : j;
                CHECK(it->second.ti == i - 1 && it->second.tj < progress_needed(G, at), "top entry of tile %d produced by tile %d: beyond progress_needed(%d)", j,
                      it->second.tj, at);
              }
            }
          }
    }
  // ---- deferred pixels ------------------------------------------------------------------------------------
  for (int t = 0; t < G.nTR; ++t)
    for (int e = 0; e < kDeferPix; ++e) {
      int x, y;
      if (!defer_pixel(G, 0, t, e, x, y)) continue;
      CHECK(defer_entry(G, 0, x, y) == e, "deferred pixel %d,%d: entry %d", x, y, e);
      const long long o = defer_in_img(G, t, e, kDeferRows - 1) + D;
      CHECK(o <= G.img_floats, "deferred block outside the image block");
    }
}

int main() {
  const int sizes[] = {1, 2, 7, 8, 9, 15, 16, 17, 40, 95, 151, 375, 450};
  int shapes = 0;
  for (int H : sizes)
    for (int W : sizes)
      for (int D : {128, 256}) {
        if ((long long)H * W > 200 * 460) continue;  // (the largest: 151 x 450, 375 x 151)
        check_shape(Shape{H, W, D});
        ++shapes;
      }
  std::printf("%d shapes, %lld checks, %lld failures\n", shapes, g_checks, g_failures);
  return g_failures ? 1 : 0;
}
