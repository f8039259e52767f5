// Host-side interface of the skewed-tile form of the semiglobal aggregation (stm_sgm_skew.hip),
// used by the form selection in stm_sgm_tiled.hip.
#pragma once

#include "stm_common.h"
#include "stm_sgm_skew_index.h"

namespace stm {

// Whether the form covers a call (D = 128 or 256, no SINGLE_PASSES / EXACT_NONFINITE / ACCUMULATE; sizes
// within the 32-bit offsets its kernels use inside one image).
bool skew_covers(int H, int W, int D, int batch, int flags);
size_t skew_workspace_bytes(int H, int W, int D, int batch);
// final_mode as in sgm_run (stm_sgm.hip): 0 sum only, 1 sum + disparity, 2 disparity only.
int skew_run(const float *cost, const float *left, float *out, int32_t *disp, int H, int W, int D,
             float penalty1, float penalty2, int flags, void *workspace, size_t workspace_bytes, int batch,
             hipStream_t stream, int final_mode, int cus);

// The two launches of a call, per disparity count (stm_sgm_skew_d128.hip / _d256.hip).
int skew_launch_d128(const float *cost, const float *left, float *out, int32_t *disp, float *ws, const skew::SkewGeom &G,
                     float p1, float p2, int dlim, int true_paths, hipStream_t stream, int final_mode, int cus);
int skew_launch_d256(const float *cost, const float *left, float *out, int32_t *disp, float *ws, const skew::SkewGeom &G,
                     float p1, float p2, int dlim, int true_paths, hipStream_t stream, int final_mode, int cus);

}  // namespace stm
