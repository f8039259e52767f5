// Semiglobal aggregation, tiled form: the horizontal walks (see stm_sgm_tiled.hip).
//
// Reference semantics: src/semiglobal.cpp:116-159, path group 1 (:24-30) and its inverse: one
// wave per image row and direction; the state is kept where the path leaves a tile
// (x % 8 == 7 going right, x % 8 == 0 going left).
#include "stm_sgm_tiled_dev.h"

namespace stm {
namespace {

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
  auto col_of = [&](int i) { return hwalk_col(W, left, i); };  // (clamped: past the end, the last column again)

  // Penalties of 64 steps at a time (as in sgm_ckpt_kernel): lane l holds step i0 + l.
  float na = image[col_of(0)], nb, pen = 0.0f;
  auto pen_inputs = [&](int i0) {
    const float carry = read_lane(na, kWave - 1);
    na = image[col_of(i0 + lane)];
    nb = lane_up(na, carry);
  };
  pen_inputs(0);

  Row<VPL> cbuf[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) cbuf[u] = load_row<VPL, true>(cost, (size_t)col_of(u), D, d0);
  Row<VPL> prev = cbuf[0];
  const int keep = left ? 0 : kT - 1;
  const int k = left ? 3 : 0;
  // The states leave through a buffer descriptor over this row's slots of the direction's
  // column lines; a step that keeps nothing stores out of range, which the hardware drops, and
  // the steps run in whole groups of PF (past the row's end: the last column's costs again,
  // nothing stored).  With `if (keeps) store` and `if (i < W)` every step ended in a join at
  // which hipcc no longer knows how many accesses are in flight: it waited for ALL of them
  // (s_waitcnt vmcnt(0)) right behind each step's load, i.e. a wave had one cost row in flight
  // instead of PF and the walk ran at the rate Little's law gives 32 waves per CU.
  // (At D = 256 a wave's loads are twice as large, the walk was at the HBM's rate already, and the
  // dropped 16-byte stores cost more than the counted waits bring: 3.6 against 3.3 ms per eight
  // 1080p pairs.  There the steps keep their conditions.)
  if constexpr (VPL == 4) {
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
          cbuf[u] = load_row<VPL, true, true>(cost, (size_t)col_of(i + PF), D, d0);
          if (i > 0) prev = sgm_step<VPL, true>(prev, c, read_lane(pen, i & (kWave - 1)), X);
          const int x = col_of(i);
          if ((x & (kT - 1)) == keep) store_row<VPL, true, true>(colb_ptr(ws, G, b, k, x >> 3, y), 0, D, d0, prev);
        }
      }
    }
    return;
  }
  constexpr unsigned kNowhere = 0x7fffffffu;
  const BufRsrc brs = make_rsrc(colb_ptr(ws, G, b, k, 0, y), colb_walk_bytes(G));
  for (int base = 0; base < W; base += PF) {
    if ((base & (kWave - 1)) == 0) {
      pen = sgm_penalty(na, nb, P.p1, P.p2);
      pen_inputs(base + kWave);
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int i = base + u;
      const Row<VPL> c = cbuf[u];
      cbuf[u] = load_row<VPL, true, true>(cost, (size_t)col_of(i + PF), D, d0);
      if (i > 0) prev = sgm_step<VPL, true>(prev, c, read_lane(pen, i & (kWave - 1)), X);
      const int x = col_of(i);
      const bool keeps = i < W && (x & (kT - 1)) == keep;
      buf_store_row<VPL>(brs, (int)(keeps ? (unsigned)d0 * 4u : kNowhere), (int)colb_walk_tc_byte(G, x >> 3), prev);
    }
  }
}

}  // namespace

int tiled_launch_hwalk(const float *cost, const float *left, float *ws, const TiledCall &c, hipStream_t s) {
  const TiledGeom &G = c.G;
  HwalkArgs A;
  A.G = G; A.p1 = c.p1; A.p2 = c.p2; A.dlim = c.dlim;
  A.only_left = c.by_rows;  // the D = 128 tile kernel then carries the left-to-right states itself
  const long waves = (A.only_left ? 1l : 2l) * G.H * G.batch;
  const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  switch (G.D / 64) {
    case 1: hipLaunchKernelGGL((sgm_hwalk_kernel<1>), grid, block, 0, s, cost, left, ws, A); break;
    case 2: hipLaunchKernelGGL((sgm_hwalk_kernel<2>), grid, block, 0, s, cost, left, ws, A); break;
    default: hipLaunchKernelGGL((sgm_hwalk_kernel<4>), grid, block, 0, s, cost, left, ws, A); break;
  }
  return check_launch("stm_sgm_f32 (tiled: horizontal walks)");
}

}  // namespace stm
