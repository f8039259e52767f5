// Semiglobal aggregation, tiled form: the cost volume is read five times and the running
// sum never travels.
//
// Reference semantics: src/semiglobal.cpp:18-91 (path table), :116-159 (recurrence),
// :191-194 (for each path: run(path); run(path.inverse())) -- see stm_sgm.hip for the
// restatement.  out(p) = ((((((L> + L<) + Ldown) + Lup) + Ldr) + Lul) + Ldl) + Lur in float,
// in that order, with the reference's quirks for the two inverse diagonal families (Q6).
//
// The paired form (stm_sgm.hip) walks every path per wave: 4 read-only checkpoint walks + 4
// pair launches, each of which reads the cost volume and read-modify-writes the sum: 56-60
// bytes per (pixel, disparity).  Here the volume is cut into 8 x 8-pixel tiles:
//
//   1. BORDER STATES.  For every direction r, L_r is stored at the pixels where a path leaves a
//      tile: rows y = 7 (mod 8) for the three downward directions, y = 0 (mod 8) for the three
//      upward ones, columns x = 7 (mod 8) for the three rightward and x = 0 (mod 8) for the
//      three leftward ones.  12/8 floats per element in all.  They come from
//        * sgm_sweep_kernel, down and up: a row sweep in which a wave owns CW image columns
//          and steps the three directions that advance one row per step (down, down-right,
//          down-left; or the three upward ones) with ONE read of each cost row.  The
//          diagonal states move one column per row: between the waves of a workgroup through
//          LDS (one barrier per row), between workgroups (column strips) through 8-byte
//          {row tag, value} granules in global memory, written write-through and polled by
//          the consumer (MI355X guide, Guideline 16, form R2).  All workgroups of a launch
//          must be resident: the grid is never larger than the CU count, and every wait is
//          bounded (a timeout sets the status word in the workspace header and the sweep
//          finishes without waiting again).
//        * sgm_hwalk_kernel, left-to-right and right-to-left: one wave per image row.
//   2. TILES.  sgm_tile_kernel (D = 256, 64) / sgm_tile_h2_kernel (D = 128): one workgroup
//      per tile loads the tile's cost rows into LDS once, rebuilds all eight directions
//      inside the tile from the border states (every path crosses at most 8 pixels of a
//      tile), adds them in the reference order into an LDS-resident sum tile and writes the
//      finished rows (and/or their winner-takes-all).
//
// At D = 128 the interior waves of the sweeps and the tile kernel hold TWO pixels per wave
// (32 lanes x 4 disparities each, stm_sgm_step.h: sgm_step_h2*): the kernels are bound by
// vector instructions issued there, and a wave-wide minimum per 128 disparities is most of a
// step's cost.
//
// Traffic per (pixel, d): 5 x 4 (cost reads) + 2 x 6 (border states written and read)
// + 4 (sum written; 0 with the fused reduction) = 32-36 bytes.
//
// Same float operations in the same order as the reference -> identical bits (tested against
// the oracle and against the paired form).  Covers D = 64, 128, 256 (whole lanes, at most
// four values per lane: the tile's cost and sum must fit 160 KB of LDS); everything else
// takes the paired form.
//
// Files: stm_sgm_tiled_index.h (geometry and every global address expression, host + device;
// checked against the buffers' extents by tests/test_tiled_index.py without a GPU),
// stm_sgm_sweep.hip, stm_sgm_hwalk.hip, stm_sgm_tile.hip, stm_sgm_tile_h2.hip (the kernels and
// their launches, stm_sgm_tiled_launch.h), stm_sgm_tile_common.h (chains, update,
// winner-takes-all of a tile), this file: plan, form selection, a call's sequence of launches.
#include "stm_sgm_tiled.h"

#include <stdlib.h>

#include "stm_sgm_skew_launch.h"
#include "stm_sgm_tiled_launch.h"

namespace stm {
namespace {

using namespace tiled;

#ifdef STM_TILE_STAMPS
unsigned long long *g_tile_stamps = nullptr;
unsigned long long *g_sweep_stamps = nullptr;
#endif

int device_cus() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  return n;
}

struct TiledPlan {
  TiledGeom G;
  long long status_off;
  size_t total_floats;
  size_t zero_bytes;  // status word + mailboxes, cleared at the start of every call
  int group;          // images per sweep launch
  int cus;
};

bool make_plan(int H, int W, int D, int batch, TiledPlan &plan) {
  const int cus = device_cus();
  if (cus <= 0) return false;
  long long total = 0, zero = 0;
  if (!make_geom(H, W, D, batch, plan.G, total, zero)) return false;
  // All strips of an image sweep together and must be resident: one workgroup per CU.
  if (plan.G.strips > cus) return false;
  plan.cus = cus;
  plan.group = cus / plan.G.strips;
  if (plan.group > batch) plan.group = batch;
  plan.status_off = 0;
  plan.total_floats = (size_t)total;
  plan.zero_bytes = (size_t)zero * sizeof(float);
  return true;
}

bool shape_ok(int H, int W, int D, int batch, int flags) {
  if (flags & (STM_SGM_SINGLE_PASSES | STM_SGM_EXACT_NONFINITE | STM_SGM_PAIRED)) return false;
  if (D != 64 && D != 128 && D != 256) return false;
  if (H < 1 || W < 1 || batch < 1 || batch > 65535) return false;
  if ((long long)batch * ((H + kT - 1) / kT) * ((W + kT - 1) / kT) > 0x3fffffffll) return false;
  return true;
}

}  // namespace

SgmForm sgm_form(int H, int W, int D, int batch, int flags) {
  if ((flags & STM_SGM_SKEW) && !(flags & STM_SGM_PAIRED) && skew_covers(H, W, D, batch, flags)) return kSgmSkew;
  // The skewed-tile form by default: large batches it covers.  Its two passes are pipelines of
  // bands (a band starts three tiles behind the one it depends on): ~5 ms for a Teddy-sized image
  // whatever the batch, so it needs several units (image, band) per workgroup to amortise that.
  // Measured on MI355X (tools/gpu_jobs/forms_sweep.sh, Teddy 375 x 450 x 128, fused, ms per pair):
  //   B =                      16     24     32     48     64     128    256
  //   tiles behind walks       0.233         0.228         0.234  0.218
  //   tiles behind sweeps      0.266  0.216  0.189  0.182  0.176  0.163  0.154
  //   skewed tiles             0.248  0.181  0.144  0.126  @64    @128   0.107
  //   1080p, D = 256           4      8      16          (tools/sgm_forms.py)
  //   tiles behind sweeps      3.88   3.63   3.64
  //   skewed tiles             4.44   3.38   2.90
  // (D = 256: one workgroup per CU, so eight 1080p pairs are 1,080 bands for 256 workgroups -- five
  // rounds for 4.2 rounds of work, and a band is a serial chain of 241 tiles that no finer unit
  // shortens: band segments were built and measured, 40.6 against 30.9 ms)
  if (!(flags & (STM_SGM_PAIRED | STM_SGM_TILED | STM_SGM_WALKS)) && skew_covers(H, W, D, batch, flags)) {
    const int cus = device_cus();
    const long long bands = (long long)batch * ((H + kT - 1) / kT), wgs = (long long)cus * (D == 128 ? 2 : 1);
    if (cus > 0 && bands >= (D == 128 ? 2 : 4) * wgs && W >= 64) return kSgmSkew;
  }
  if (!shape_ok(H, W, D, batch, flags)) return kSgmPaired;
  TiledGeom G;
  long long total = 0, zero = 0;
  if (!make_geom(H, W, D, batch, G, total, zero)) return kSgmPaired;
  // The walk-fed variant needs nothing of the device; the sweep-fed one needs its strips on
  // distinct CUs (make_plan) and a sweep workgroup that fits one (tiled_sweep_fits).
  TiledPlan plan;
  const bool sweeps_ok = make_plan(H, W, D, batch, plan) && tiled_sweep_fits(D);
  if (flags & STM_SGM_WALKS) return kSgmTiledWalks;
  if (flags & STM_SGM_TILED) return sweeps_ok ? kSgmTiledSweeps : kSgmTiledWalks;
  if (D < 128) return kSgmPaired;  // D = 64: the paired form is 1.4x faster than either
  // Measured on MI355X (tools/sgm_forms.py, fused, ms per pair):
  //   Teddy 375 x 450, D = 128     B = 1     2     4     16    24    32    256
  //     paired                     0.570 0.346 0.263 0.236 0.229 0.230 0.215
  //     tiles behind walks         0.307 0.227 0.234 0.229 0.228 0.227
  //     tiles behind sweeps        3.53        0.93  0.29  0.23  0.19  0.162
  //   1080p, D = 256               B = 1     2     3     8
  //     paired                     5.89  5.98  5.75  5.63
  //     tiles behind walks         5.25  5.22  5.00
  //     tiles behind sweeps        7.96  5.34  4.5   4.0
  // (240 x 320 x 128: 0.196 against 0.379; 96 x 128 x 128: 0.081 against 0.170; 64 x 64 x 256: 0.046
  // against 0.120 for one pair.)  A sweep costs its image rows x ~3 us whatever the batch, so the
  // sweep-fed form needs enough columns in the batch to keep its column strips on most CUs;
  // below that the walk-fed tiles are never behind the paired form.
  if (sweeps_ok && H >= 64) {
    const long long need = (long long)plan.cus * sweep_strip_cols(D / 64) * (D == 256 ? 12 : 15);  // tenths
    if ((long long)batch * W * 2 * 10 >= need) return kSgmTiledSweeps;
  }
  return (H >= 16 && W >= 16) ? kSgmTiledWalks : kSgmPaired;
}

bool sgm_tiled_aligned(const void *cost, const void *out, const void *workspace, int final_mode) {
  (void)final_mode;  // (`out` is read by STM_SGM_ACCUMULATE even when it is not written)
  return (reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && (reinterpret_cast<uintptr_t>(cost) & 15) == 0 &&
         (reinterpret_cast<uintptr_t>(out) & 15) == 0;
}

size_t sgm_tiled_workspace_bytes(int H, int W, int D, int batch, int flags) {
  if (sgm_form(H, W, D, batch, flags) == kSgmPaired) return 0;
  if (sgm_form(H, W, D, batch, flags) == kSgmSkew) return skew_workspace_bytes(H, W, D, batch);
  TiledGeom G;
  long long total = 0, zero = 0;
  make_geom(H, W, D, batch, G, total, zero);
  return (size_t)total * sizeof(float);
}

int sgm_run_tiled(const float *cost, const float *left, float *out, int32_t *disp, int H, int W,
                  int D, float penalty1, float penalty2, int flags, void *workspace,
                  size_t workspace_bytes, int batch, hipStream_t stream, int final_mode) {
  const SgmForm form = sgm_form(H, W, D, batch, flags);
  STM_REQUIRE(form != kSgmPaired, STM_EUNSUPPORTED, "stm_sgm_f32: the tiled form does not cover this call");
  if (form == kSgmSkew) {
    STM_REQUIRE(sgm_tiled_aligned(cost, out, workspace, final_mode), STM_EINVAL,
                "stm_sgm_f32: the tiled forms need 16-byte aligned volumes and workspace");
    const int cus = device_cus();
    STM_REQUIRE(cus > 0, STM_EHIP, "stm_sgm_f32: no HIP device");
    return skew_run(cost, left, out, disp, H, W, D, penalty1, penalty2, flags, workspace, workspace_bytes, batch,
                    stream, final_mode, cus);
  }
  TiledPlan plan;
  if (form == kSgmTiledSweeps) {
    make_plan(H, W, D, batch, plan);
  } else {
    long long total = 0, zero = 0;
    make_geom(H, W, D, batch, plan.G, total, zero);
    plan.total_floats = (size_t)total;
    plan.zero_bytes = 0;  // (no mailboxes; the walk kernel clears the status word itself)
    plan.status_off = 0;
    plan.group = 0;
    plan.cus = device_cus();
    STM_REQUIRE(plan.cus > 0, STM_EHIP, "stm_sgm_f32: no HIP device");
  }
  const size_t need = plan.total_floats * sizeof(float);
  STM_REQUIRE(workspace && workspace_bytes >= need, STM_EWORKSPACE,
              "stm_sgm_f32: needs a workspace of %zu bytes (stm_sgm_workspace_bytes)", need);
  STM_REQUIRE(sgm_tiled_aligned(cost, out, workspace, final_mode), STM_EINVAL,
              "stm_sgm_f32: the tiled form needs 16-byte aligned volumes and workspace");
  static_assert(kHeaderFloats * sizeof(float) == kSgmWorkspaceHeader, "workspace header");
  TiledCall c;
  c.G = plan.G;
  c.p1 = penalty1; c.p2 = penalty2;
  c.dlim = (flags & STM_SGM_FULL_MIN) ? D : D - 1;
  c.true_paths = (flags & STM_SGM_TRUE_PATHS) ? 1 : 0;
  c.accumulate = (flags & STM_SGM_ACCUMULATE) ? 1 : 0;
  c.cooperative = (flags & STM_SGM_COOPERATIVE) != 0;
  c.by_rows = form == kSgmTiledSweeps;
  // Segments per tile row (D = 64, 256 behind sweeps; stm_sgm_tiled_index.h, row_segs): finer units
  // balance better -- eight 1080p pairs are 1,080 tile rows for 256 workgroups -- and every
  // segment start costs a flag wait and a cold entry load: at least sixteen tiles per segment.
  // Measured, eight 1080p pairs: 32.4 / 30.7 / 29.6 / 29.6 ms with 1 / 2 / 3 / 4 segments, 32.7 with
  // the left-to-right direction walked instead of carried.
  c.segs = plan.G.nTC / 16 < 1 ? 1 : (plan.G.nTC / 16 > kMaxSegs ? kMaxSegs : plan.G.nTC / 16);
#ifdef STM_DEV_SWITCHES  // (development builds only, tools/build_probe.sh: the product reads no environment)
  if (getenv("STM_TILE_SEGS")) c.segs = atoi(getenv("STM_TILE_SEGS"));
  if (getenv("STM_TILE_NO_ROWS") && D != 128) c.by_rows = false;  // (A/B against the walked form)
#endif
  c.segs = c.segs < 1 ? 1 : (c.segs > kMaxSegs ? kMaxSegs : c.segs);  // (the flag area holds kMaxSegs per tile row)
  c.status_off = plan.status_off;
  c.group = plan.group;
  c.cus = plan.cus;
  c.sweep_stamps = c.tile_stamps = nullptr;
#ifdef STM_TILE_STAMPS
  c.sweep_stamps = g_sweep_stamps;
  c.tile_stamps = g_tile_stamps;
  // diagnostic build only: leave out the walks (1), the sweeps (2), the tiles (4)
  const int skip = getenv("STM_TILED_SKIP") ? atoi(getenv("STM_TILED_SKIP")) : 0;
#else
  constexpr int skip = 0;
#endif
  float *ws = static_cast<float *>(workspace);
  if (plan.zero_bytes && hipMemsetAsync(ws, 0, plan.zero_bytes, stream) != hipSuccess)
    return fail(STM_EHIP, "stm_sgm_f32 (tiled): hipMemsetAsync failed");
  int rc = STM_OK;
  if (form == kSgmTiledWalks) {
    if (!(skip & 3)) rc = tiled_launch_bwalk(cost, left, ws, c, stream);
  } else {
    if (!(skip & 1)) rc = tiled_launch_hwalk(cost, left, ws, c, stream);
    for (int up = 0; up < 2 && rc == STM_OK && !(skip & 2); ++up)
      rc = tiled_launch_sweep(cost, left, ws, c, up != 0, stream);
  }
  if (rc != STM_OK || (skip & 4)) return rc;
  return D == 128 ? tiled_launch_tiles_h2(cost, left, out, disp, ws, c, final_mode, stream)
                  : tiled_launch_tiles(cost, left, out, disp, ws, c, final_mode, stream);
}

}  // namespace stm

#ifdef STM_TILE_STAMPS
extern "C" void stm_debug_set_tile_stamps(void *p) {
  stm::g_tile_stamps = static_cast<unsigned long long *>(p);
}
extern "C" void stm_debug_set_sweep_stamps(void *p) {
  stm::g_sweep_stamps = static_cast<unsigned long long *>(p);
}
#endif
