// Semiglobal aggregation, tiled form fed by WALKS instead of row sweeps (see stm_sgm_tiled.hip).
//
// Reference semantics: src/semiglobal.cpp:116-159 (recurrence) along the eight directions of
// :18-91.  One wave per path and direction reads only the cost volume and keeps the state where
// the path leaves an 8 x 8 tile (stm_sgm_tiled_index.h: bwalk_line, bwalk_stores); the tile
// kernels then rebuild every direction inside the tiles exactly as behind the row sweeps.
//
// Why a second way to the same border states: a row sweep costs its image rows x ~3 us of
// serial hand-offs whatever the number of images, and needs every workgroup of a launch
// resident.  Walks have neither -- a path is a wave, thousands run side by side -- at the price
// of reading the cost volume once per direction (8, or 7 at D = 128) instead of once per three:
// the form of choice for the reference's call shape, ONE pair per call (pipeline.py:69-94), and
// for batches too small to fill the chip with sweep strips.  No waits, no status word to set.
#include "stm_sgm_tiled_dev.h"

namespace stm {
namespace {

struct BwalkArgs {
  TiledGeom G;
  float p1, p2;
  int dlim;
  int skip_right;  // D = 128: the tile kernel carries the left-to-right states along its tile rows
  int true_paths;  // STM_SGM_TRUE_PATHS: every inverse diagonal is needed
};

template <int VPL>
__global__ __launch_bounds__(256) void sgm_bwalk_kernel(const float *__restrict__ cost_all,
                                                        const float *__restrict__ image_all,
                                                        float *__restrict__ ws, const BwalkArgs P) {
  constexpr int PF = 8;
  constexpr int D = 64 * VPL;
  const TiledGeom &G = P.G;
  const int lane = threadIdx.x & 63;
  const int wave = wave_uniform((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int per_image = bwalk_lines(G, P.skip_right != 0);
  if (wave >= per_image * G.batch) return;
  // The workspace's status word: this variant has no waits that could expire (one launch less
  // than a memset node, which counts for a single pair).
  if (wave == 0 && lane == 0) *reinterpret_cast<unsigned *>(ws) = 0u;
  // Lines in the outer position, images in the inner one: the long lines (rows, columns) of
  // every image start first and the short diagonals fill in behind them.
  const int l = wave / G.batch, b = wave - l * G.batch;
  const WalkLine k = bwalk_line(G, P.skip_right != 0, l);
  if (!bwalk_line_used(G, k.dir, k.index, P.true_paths)) return;  // (the reference never runs this pass)
  const WalkStores st = bwalk_stores(k.dir);
  const int d0 = lane * VPL;
  const size_t npix = (size_t)G.H * G.W;
  const float *cost = cost_all + (size_t)b * npix * D;
  const float *image = image_all + (size_t)b * npix;
  StepCtx X;
  X.init(P.p1, d0, P.dlim, lane == 63 && P.dlim < D);

  // Penalties of 64 steps at a time (as in sgm_hwalk_kernel): lane l holds step i0 + l.
  float na = image[bwalk_pixel(G, k, 0)], nb, pen = 0.0f;
  auto pen_inputs = [&](int i0) {
    const float carry = read_lane(na, kWave - 1);
    na = image[bwalk_pixel(G, k, i0 + lane)];
    nb = lane_up(na, carry);
  };
  pen_inputs(0);

  Row<VPL> cbuf[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) cbuf[u] = load_row<VPL, true>(cost, (size_t)bwalk_pixel(G, k, u), D, d0);
  Row<VPL> prev = cbuf[0];
  int x = k.x0, y = k.y0;
  for (int base = 0; base < k.len; base += PF) {
    if ((base & (kWave - 1)) == 0) {
      pen = sgm_penalty(na, nb, P.p1, P.p2);
      pen_inputs(base + kWave);
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int i = base + u;
      if (i < k.len) {
        const Row<VPL> c = cbuf[u];
        cbuf[u] = load_row<VPL, true>(cost, (size_t)bwalk_pixel(G, k, i + PF), D, d0);
        if (i > 0) prev = sgm_step<VPL, true>(prev, c, read_lane(pen, i & (kWave - 1)), X);
        if (st.rk >= 0 && (y & (kT - 1)) == st.rv)
          store_row<VPL, true>(rowb_ptr(ws, G, b, st.rk, y >> 3, x), 0, D, d0, prev);
        if (st.ck >= 0 && (x & (kT - 1)) == st.cv)
          store_row<VPL, true>(colb_ptr(ws, G, b, st.ck, x >> 3, y), 0, D, d0, prev);
        x += k.dx;
        y += k.dy;
      }
    }
  }
}

}  // namespace

int tiled_launch_bwalk(const float *cost, const float *left, float *ws, const TiledCall &c, hipStream_t s) {
  const TiledGeom &G = c.G;
  BwalkArgs A;
  A.G = G; A.p1 = c.p1; A.p2 = c.p2; A.dlim = c.dlim;
  A.skip_right = c.by_rows;
  A.true_paths = c.true_paths;
  const long long waves = (long long)bwalk_lines(G, A.skip_right != 0) * G.batch;
  const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  switch (G.D / 64) {
    case 1: hipLaunchKernelGGL((sgm_bwalk_kernel<1>), grid, block, 0, s, cost, left, ws, A); break;
    case 2: hipLaunchKernelGGL((sgm_bwalk_kernel<2>), grid, block, 0, s, cost, left, ws, A); break;
    default: hipLaunchKernelGGL((sgm_bwalk_kernel<4>), grid, block, 0, s, cost, left, ws, A); break;
  }
  return check_launch("stm_sgm_f32 (tiled: border walks)");
}

}  // namespace stm
