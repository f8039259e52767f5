// Semiglobal aggregation, tiled form: the tile kernel for D = 64 and D = 256, one line of a
// direction pair per wave (see stm_sgm_tiled.hip for the whole form, stm_sgm_tile_common.h for
// the chains and the update).
#include "stm_sgm_tile_common.h"

namespace stm {
namespace {

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
  const int w16 = wave_uniform((int)(threadIdx.x >> 6));  // (also the stamps' wave id)
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
      // (rows and columns outside a partial tile are clamped into it: tile_cost_pixel)
#pragma unroll
      for (int c = 0; c < kT; ++c)
        cpre[c] = load_row_at<VPL, true>(cost, (long)(tile_cost_pixel(G, x0, y0, tw, th, w, c) * (unsigned)D), D, d0);
      // Intensities of the tile and the ring of pixels around it (clamped at the image border;
      // a clamped sample is only ever used for a step that does not exist).
      ipre = image_all[(size_t)t.b * npix +
                       tile_intensity_pixel(G, x0, y0, imin((int)threadIdx.x, (kT + 2) * (kT + 2) - 1))];
      load_entries<VPL, 0>(E0, tile_geom(0, w, tw, th), border_base(ws, G, t.b), G, x0, y0, d0);
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
      STM_STAMP_TILE_ID(id);
      STM_STAMP(0);
      const Entries<VPL, 0> E0now = E0;
      // (the previous tile's winner-takes-all has read the sum tile: both groups are past B4)
      if (P.accumulate) stage_accumulate_row<VPL>(St, out, G, x0, y0, tw, th, w, d0);
      STM_STAMP(1);
      lds_barrier();  // B0
      STM_STAMP(2);
      tile_pair<VPL, 0, FINAL>(tile_geom(0, w, tw, th), E0now, full, Ct, St, Ip, P, x0, y0, out, disp, X, lane,
                               [] {});
      STM_STAMP(3);
      const PhaseGeom g2 = tile_geom(2, w, tw, th);
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
        tile_wta_lds<D, 2 * kT>(St, w16, lane, tw, th, G, x0, y0, disp);
      }
      cur_t = next_t;
      STM_STAMP(7);
    }
  } else {
    Entries<VPL, 1> E1;  // next tile: border states of pair 1
    auto prefetch = [&](const TileWalk &t) {
      const int x0 = t.tcol * kT, y0 = t.trow * kT;
      const int tw = imin(kT, G.W - x0), th = imin(kT, G.H - y0);
      load_entries<VPL, 1>(E1, tile_geom(1, w, tw, th), border_base(ws, G, t.b), G, x0, y0, d0);
    };
    prefetch(next_t);
    for (; id < ntiles; id += stride) {
      const int b = cur_t.b, x0 = cur_t.tcol * kT, y0 = cur_t.trow * kT;
      const int tw = imin(kT, G.W - x0), th = imin(kT, G.H - y0);
      const bool full = tw == kT && th == kT;
      float *out = out_all ? out_all + (size_t)b * npix * D : nullptr;
      int32_t *disp = disp_all ? disp_all + (size_t)b * npix : nullptr;
      STM_STAMP_TILE_ID(id);
      STM_STAMP(0);
      const Entries<VPL, 1> E1now = E1;
      STM_STAMP(1);
      lds_barrier();  // B0
      STM_STAMP(2);
      tile_pair<VPL, 1, FINAL>(tile_geom(1, w, tw, th), E1now, full, Ct, St, Ip, P, x0, y0, out, disp, X, lane,
                               [&] {
                                 STM_STAMP(3);
                                 lds_barrier();  // B1
                               });
      const PhaseGeom g3 = tile_geom(3, w, tw, th);
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
        tile_wta_lds<D, 2 * kT>(St, w16, lane, tw, th, G, x0, y0, disp);
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

}  // namespace

int tiled_launch_tiles(const float *cost, const float *left, float *out, int32_t *disp, float *ws,
                       const TiledCall &c, int final_mode, hipStream_t s) {
  const TiledGeom &G = c.G;
  TileArgs A;
  A.G = G; A.p1 = c.p1; A.p2 = c.p2; A.dlim = c.dlim; A.true_paths = c.true_paths; A.accumulate = c.accumulate;
  A.stamps = c.tile_stamps;
  A.by_rows = 0;
  // Persistent workgroups; tiles are dealt round robin.  D = 256: one sixteen-wave workgroup per
  // CU (the LDS allows no more); D = 64: two.
  const long long ntiles = (long long)G.nTR * G.nTC * G.batch;
  const long long wgs = (long long)c.cus * (G.D == 256 ? 1 : 2);
  const dim3 grid((unsigned)(ntiles < wgs ? ntiles : wgs)), block(2 * kT * 64);
#define STM_TILE(VPL_)                                                                                      \
  do {                                                                                                      \
    if (final_mode == 0)                                                                                    \
      hipLaunchKernelGGL((sgm_tile_kernel<VPL_, 0>), grid, block, 0, s, cost, left, out, disp, ws, A);      \
    else if (final_mode == 1)                                                                               \
      hipLaunchKernelGGL((sgm_tile_kernel<VPL_, 1>), grid, block, 0, s, cost, left, out, disp, ws, A);      \
    else                                                                                                    \
      hipLaunchKernelGGL((sgm_tile_kernel<VPL_, 2>), grid, block, 0, s, cost, left, out, disp, ws, A);      \
  } while (0)
  if (G.D == 256) STM_TILE(4);
  else STM_TILE(1);
#undef STM_TILE
  return check_launch("stm_sgm_f32 (tiled: tiles)");
}

}  // namespace stm
