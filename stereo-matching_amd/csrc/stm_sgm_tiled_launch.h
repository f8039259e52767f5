// Interface between the translation units of the tiled form: what stm_sgm_tiled.hip (plan,
// form selection, the call's sequence of launches) needs from the kernels' files.
#pragma once

#include "stm_common.h"
#include "stm_sgm_tiled_index.h"

namespace stm {

// One call of the tiled form.
struct TiledCall {
  tiled::TiledGeom G;
  float p1, p2;
  int dlim;           // disparities the per-step minimum runs over (D - 1: reference, D: STM_SGM_FULL_MIN)
  int true_paths;     // STM_SGM_TRUE_PATHS
  int accumulate;     // STM_SGM_ACCUMULATE
  bool cooperative;   // STM_SGM_COOPERATIVE
  bool by_rows;       // behind sweeps: the tile kernel owns tile rows (D = 128) / row segments and carries
                      // the left-to-right state itself (no walk of that direction)
  int segs;           // D = 64, 256: segments per tile row
  long long status_off;
  int group;          // images per sweep launch
  int cus;
  unsigned long long *sweep_stamps, *tile_stamps;  // diagnostic builds only (stm_sgm_tiled_diag.h), else null
};

// stm_sgm_hwalk.hip: the horizontal walks (both directions; D = 128: right-to-left only).
int tiled_launch_hwalk(const float *cost, const float *left, float *ws, const TiledCall &c, hipStream_t s);
// stm_sgm_bwalk.hip: the border states of all eight directions (seven at D = 128) by walks, in
// place of the horizontal walks and the two sweeps.
int tiled_launch_bwalk(const float *cost, const float *left, float *ws, const TiledCall &c, hipStream_t s);
// stm_sgm_sweep.hip: the down (up = false) or up row sweep over the whole batch.
int tiled_launch_sweep(const float *cost, const float *left, float *ws, const TiledCall &c, bool up, hipStream_t s);
// Whether one sweep workgroup fits a CU of the current device at this D (asked once per device).
bool tiled_sweep_fits(int D);
// stm_sgm_tile.hip (D = 64, 256) / stm_sgm_tile_h2.hip (D = 128): the tiles.  final_mode: 0 sum
// only, 1 sum + disparity, 2 disparity only.
int tiled_launch_tiles(const float *cost, const float *left, float *out, int32_t *disp, float *ws,
                       const TiledCall &c, int final_mode, hipStream_t s);
int tiled_launch_tiles_h2(const float *cost, const float *left, float *out, int32_t *disp, float *ws,
                          const TiledCall &c, int final_mode, hipStream_t s);

}  // namespace stm
