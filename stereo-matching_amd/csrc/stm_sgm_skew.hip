// Semiglobal aggregation, SKEWED-TILE form (D = 128, 256): two passes over sheared 8 x 8 tiles, each
// reading the cost volume once; the four directions of a pass are carried from tile to tile
// (registers inside a band, row border states + progress counters between bands).  Geometry,
// chains and the reasoning: stm_sgm_skew_index.h.
//
// Reference semantics: src/semiglobal.cpp:18-91 (path table), :116-159 (recurrence), :191-194
// (for each path: run(path); run(path.inverse())): out(p) = ((((((L> + L<) + Ldown) + Lup) + Ldr)
// + Lul) + Ldl) + Lur in float, in that order, with the reference's quirks for the two inverse
// diagonal families (Q6: line_counts).  Same float operations in the same order as the other
// forms -> identical bits (tests/test_gpu_sgm_skew.py: against the oracle and the paired form).
//
// D = 128: two pixels per wave (stm_sgm_step.h, sgm_step_h2*): lanes 0-31 one pixel's 128
// disparities, lanes 32-63 another's; a half-wave owns one CHAIN (8 slots) per pair, four waves
// per tile.  D = 256: one pixel per wave, a wave owns one chain per pair, eight waves per tile.
// The kernels (stm_sgm_skew_body.h) are compiled once per D: stm_sgm_skew_d128.hip, _d256.hip.
//
//   sgm_skew_rev_kernel  bands bottom-up, tiles right-to-left; left / up / up-left / up-right.
//                        LDS: the cost tile (32 KB): four workgroups per CU.
//   sgm_skew_fwd_kernel  bands top-down, tiles left-to-right; carries right / down / down-right /
//                        down-left, rebuilds the other four from the reverse pass's states, sum
//                        tile in LDS (36 KB), winner-takes-all.  Two workgroups per CU.
//
// Inter-workgroup protocol (MI355X guide, Guideline 16 / "Valid forms"): a band's owner stores
// the states that leave its tiles downwards (upwards) with sc1 stores; every wave drains its
// stores (s_waitcnt vmcnt(0)) at the top of the next tile, the workgroup barrier follows, then
// ONE lane publishes the band's progress counter with an sc1 store.  The owner of the next band
// polls that counter with sc1 loads (one wave), joins its workgroup's barrier, and only then
// are the states loaded, with sc1 loads.  Units (image, band) are taken by an atomic ticket in
// the order "band 0 of every image, band 1 of every image, ...": the band a unit waits for has
// a lower ticket, so it is held by a workgroup that runs or is done -- no residency requirement,
// no deadlock.  Waits are bounded (~2 s -> status word), like the sweeps'.
#include "stm_sgm_skew_launch.h"  // (stm_common.h first: the index header needs the HIP runtime's qualifiers)

namespace stm {

bool skew_covers(int H, int W, int D, int batch, int flags) {
  if (flags & (STM_SGM_SINGLE_PASSES | STM_SGM_EXACT_NONFINITE | STM_SGM_ACCUMULATE)) return false;
  if (D != 128 && D != 256) return false;
  if (H < 1 || W < 1 || batch < 1 || batch > 65535) return false;
  skew::SkewGeom G;
  long long total = 0, zero = 0;
  if (!skew::make_geom(H, W, D, batch, G, total, zero)) return false;
  return (long long)batch * G.nTR < 0x3fffffffll;
}

size_t skew_workspace_bytes(int H, int W, int D, int batch) {
  skew::SkewGeom G;
  long long total = 0, zero = 0;
  skew::make_geom(H, W, D, batch, G, total, zero);
  return (size_t)total * sizeof(float);
}

int skew_run(const float *cost, const float *left, float *out, int32_t *disp, int H, int W, int D,
             float penalty1, float penalty2, int flags, void *workspace, size_t workspace_bytes, int batch,
             hipStream_t stream, int final_mode, int cus) {
  skew::SkewGeom G;
  long long total = 0, zero = 0;
  skew::make_geom(H, W, D, batch, G, total, zero);
  const size_t need = (size_t)total * sizeof(float);
  STM_REQUIRE(workspace && workspace_bytes >= need, STM_EWORKSPACE,
              "stm_sgm_f32: needs a workspace of %zu bytes (stm_sgm_workspace_bytes)", need);
  float *ws = static_cast<float *>(workspace);
  // status word, unit tickets, progress counters
  if (hipMemsetAsync(ws, 0, (size_t)zero * sizeof(float), stream) != hipSuccess)
    return fail(STM_EHIP, "stm_sgm_f32 (skewed tiles): hipMemsetAsync failed");
  const int dlim = (flags & STM_SGM_FULL_MIN) ? D : D - 1;
  const int true_paths = (flags & STM_SGM_TRUE_PATHS) ? 1 : 0;
  return D == 128 ? skew_launch_d128(cost, left, out, disp, ws, G, penalty1, penalty2, dlim, true_paths, stream,
                                     final_mode, cus)
                  : skew_launch_d256(cost, left, out, disp, ws, G, penalty1, penalty2, dlim, true_paths, stream,
                                     final_mode, cus);
}

}  // namespace stm
