// Host-side interface of the tiled form of the semiglobal aggregation (stm_sgm_tiled.hip),
// used by the C entry points in stm_sgm.hip.
#pragma once

#include "stm_common.h"

namespace stm {

// Which form the aggregation takes for this call (shape, flags, device): the paired form
// (stm_sgm.hip), or the tiled form with its border states from row sweeps (large batches; the
// workspace's status word can be set) or from per-path walks (single pairs and small batches;
// no waits, the status word stays 0).  Deterministic in its arguments and the current device, so
// that stm_sgm_workspace_bytes() and stm_sgm_f32() always agree.
enum SgmForm { kSgmPaired = 0, kSgmTiledSweeps = 1, kSgmTiledWalks = 2, kSgmSkew = 3 };
SgmForm sgm_form(int H, int W, int D, int batch, int flags);
inline bool sgm_tiled_selected(int H, int W, int D, int batch, int flags) {
  return sgm_form(H, W, D, batch, flags) != kSgmPaired;
}

// The tiled form moves whole 16-byte lanes: cost, out and workspace must be 16-byte aligned
// (a call that is not, and does not force STM_SGM_TILED, takes the paired form instead).
bool sgm_tiled_aligned(const void *cost, const void *out, const void *workspace, int final_mode);

// Bytes at the head of the workspace of every call for which sgm_tiled_selected() holds: the
// status word (whichever form then runs), padded so that what follows stays 16-byte aligned.
constexpr size_t kSgmWorkspaceHeader = 256;

// Workspace the tiled form needs for this call (0 when it is not selected).
size_t sgm_tiled_workspace_bytes(int H, int W, int D, int batch, int flags);

// final_mode as in sgm_run (stm_sgm.hip): 0 sum only, 1 sum + disparity, 2 disparity only.
// Runs whichever tiled variant sgm_form() names for the call.
int sgm_run_tiled(const float *cost, const float *left, float *out, int32_t *disp, int H, int W,
                  int D, float penalty1, float penalty2, int flags, void *workspace,
                  size_t workspace_bytes, int batch, hipStream_t stream, int final_mode);

}  // namespace stm
