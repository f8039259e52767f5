// Device-side pieces shared by the kernels of the tiled form of the semiglobal aggregation
// (stm_sgm_sweep.hip, stm_sgm_hwalk.hip, stm_sgm_tile.hip, stm_sgm_tile_h2.hip); the geometry
// and every global address expression live in stm_sgm_tiled_index.h.
#pragma once

#include "stm_sgm_step.h"
#include "stm_sgm_tiled_diag.h"
#include "stm_sgm_tiled_index.h"
#include "stm_sgm_tiled_launch.h"

namespace stm {
namespace {

using namespace tiled;

typedef unsigned long long u64;

__device__ __forceinline__ float *rowb_ptr(float *ws, const TiledGeom &G, int b, int k, int t, int x) {
  return ws + rowb_off(G, b, k, t, x);
}
__device__ __forceinline__ float *colb_ptr(float *ws, const TiledGeom &G, int b, int k, int tc, int y) {
  return ws + colb_off(G, b, k, tc, y);
}


// a or b, element by element (a conditional on the structs would select between their
// addresses and put both in scratch memory).
template <int VPL>
__device__ __forceinline__ Row<VPL> pick_row(bool take_a, const Row<VPL> &a, const Row<VPL> &b) {
  Row<VPL> r;
#pragma unroll
  for (int j = 0; j < VPL; ++j) r.v[j] = take_a ? a.v[j] : b.v[j];
  return r;
}

}  // namespace
}  // namespace stm
