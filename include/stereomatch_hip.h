/*
 * stereomatch_hip.h -- C ABI of libstereomatch_hip.so, the MI355X (gfx950)
 * implementation of the stereo cost-volume hot path.
 *
 * Each entry point replaces one static method of the reference's pybind11
 * extension `stereomatch._cstereomatch` (upstream src/_cstereomatch.cpp:10-17):
 *
 *   stm_ssd_f32         <- CostOps.compute_ssd            src/cost.cpp:6-9,  include/stereomatch/cost.hpp:10-13, src/ssd.cu:64-81
 *   stm_birchfield_f32  <- CostOps.compute_birchfield     src/cost.cpp:14,   include/stereomatch/cost.hpp:19-22, src/birchfield_cost.cu:153-181
 *   stm_sgm_f32         <- AggregationOps.run_semiglobal  src/aggregation.cpp:6, include/stereomatch/aggregation.hpp:8-10, src/semiglobal.cpp:167-197
 *   stm_wta_f32         <- DisparityReduceOps.run_winners_take_all    src/disparity_reduce.cpp:6-7, include/stereomatch/disparity_reduce.hpp:8-9, src/winners_take_all.cu:89-114
 *   stm_dp_f32          <- DisparityReduceOps.run_dynamic_programming src/disparity_reduce.cpp:8-9, include/stereomatch/disparity_reduce.hpp:10-13, src/dynamic_programming.cu:184-225
 *
 * Conventions
 *  - Plain pointers and sizes only; no torch types.  All pointers are DEVICE
 *    pointers on the current HIP device unless stated otherwise.
 *  - Images are [batch][H][W] float32, volumes [batch][H][W][D] float32 with D
 *    innermost, disparity maps [batch][H][W] int32, all contiguous.  batch >= 1
 *    processes independent stereo pairs in one launch sequence (the reference
 *    has no batch axis; batch == 1 is its exact call shape).
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Work is
 *    enqueued stream-ordered; no entry point synchronises the device,
 *    allocates device memory or copies to/from the host, so all of them can be
 *    captured into a hipGraph.
 *  - The library keeps no state between calls.  Scratch memory, where needed,
 *    is supplied by the caller (`workspace`, sized by *_workspace_bytes).
 *  - Return value: STM_OK or an STM_E* code; stm_last_error() then returns a
 *    thread-local, human-readable message (the Python wrappers raise
 *    RuntimeError with it, matching the reference's std::runtime_error ->
 *    RuntimeError convention, include/stereomatch/check.hpp:12-19).
 *  - Results are the reference CPU path's, bit for bit (see DESIGN.md for the
 *    quirks this includes); floating-point contraction is disabled in the
 *    device code.
 */
#ifndef STEREOMATCH_HIP_H_
#define STEREOMATCH_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STM_ABI_VERSION 1

#define STM_OK 0
#define STM_EINVAL 1       /* bad argument (null pointer, non-positive size, misalignment) */
#define STM_EUNSUPPORTED 2 /* valid but outside what the kernels cover (e.g. D > 1024) */
#define STM_EHIP 3         /* a HIP runtime call or kernel launch failed */
#define STM_EWORKSPACE 4   /* workspace missing or too small */

/* Largest disparity range the wave-per-path / wave-per-pixel kernels cover
 * (64 lanes x 16 values).  The reference's CUDA path stops at 1024 as well
 * (one thread per disparity, src/semiglobal_gpu.cu:109-120). */
#define STM_MAX_DISPARITY 1024

int stm_abi_version(void);
const char *stm_last_error(void);

/* ---- SSD cost volume (src/ssd.cu:31-61) ------------------------------------
 * cost[r][c][d] = sum over rows [max(r-k,0), min(r+k,H)) and columns
 * [max(c-k,d), min(c+k,W)) of (L - R shifted by d)^2 in row-major order with
 * separate multiply and add; +inf for d > c.  k = kernel_size (reference
 * default 7).
 *
 * mode:
 *   STM_SSD_ORDERED  every output is the reference's ordered 4k^2-tap sum:
 *                    bit-exact for any input.
 *   STM_SSD_BOX      separable box sums (vertical then sliding horizontal).
 *                    Bit-exact only when every partial sum is an exactly
 *                    representable integer, i.e. integer-valued images with
 *                    (2k)^2 * (value range)^2 <= 2^24 (e.g. 8-bit images, k<=8).
 *   STM_SSD_AUTO     a device-side classifier checks that condition on the
 *                    actual images and picks BOX when it holds, ORDERED
 *                    otherwise; results are bit-exact either way.  Needs a
 *                    workspace of stm_ssd_workspace_bytes().
 */
#define STM_SSD_AUTO 0
#define STM_SSD_ORDERED 1
#define STM_SSD_BOX 2
size_t stm_ssd_workspace_bytes(int H, int W, int D, int kernel_size, int batch);
int stm_ssd_f32(const float *left, const float *right, float *cost, int H,
                int W, int D, int kernel_size, int mode, void *workspace,
                size_t workspace_bytes, int batch, void *stream);

/* ---- Birchfield-Tomasi cost (src/birchfield_cost.cu:26-67) -------------------
 * 8-tap window (the reference hard-codes kKernelSize = 4 and ignores the
 * kernel_size argument, :167,175).  Follows the CPU functor including its
 * +1-shifted sample indexing; flat indices past the end of an image read as
 * 0.0f (the reference reads out of bounds there).  Entries d > x are left
 * untouched, as in the reference.
 */
int stm_birchfield_f32(const float *left, const float *right, float *cost,
                       int H, int W, int D, int batch, void *stream);
/* Same with options (extension).  STM_BT_PADDED: the sample indexing of the reference's
 * CUDA functor (src/birchfield_cost.cu:92-139: centre sample p, neighbours p-1 / p+1, zero
 * outside the row) instead of the CPU functor's shifted flat indexing. */
#define STM_BT_PADDED 1
int stm_birchfield_ex_f32(const float *left, const float *right, float *cost,
                          int H, int W, int D, int flags, int batch,
                          void *stream);

/* ---- Semiglobal aggregation (src/semiglobal.cpp:18-160,167-197) ----------------
 * out = sum over the reference's path passes of L_r, accumulated in the
 * reference's path-table order (8 directions, including its quirks for the
 * two inverse-diagonal families and the min-over-[0,D-2] term).
 *
 * flags:
 *   STM_SGM_ACCUMULATE  out += ... on top of the caller's contents (the
 *                       reference semantics on a pre-zeroed tensor).  Without
 *                       it `out` is overwritten, which is identical to
 *                       accumulating into zeros and saves one read of `out`.
 *                       Both forms (paired, tiled) implement it.
 */
#define STM_SGM_ACCUMULATE 1
/*   STM_SGM_SINGLE_PASSES  run the eight direction families as eight separate passes over
 *                       the sum (92N bytes of traffic) instead of the default paired
 *                       form (a direction and its inverse share one read-modify-write;
 *                       ~62N bytes, needs a larger workspace).  Same results bit for bit;
 *                       kept for cross-checking and for callers short of memory. */
#define STM_SGM_SINGLE_PASSES 2
/* Corrected-algorithm options (extensions; the default is the reference's behaviour):
 *   STM_SGM_FULL_MIN    the per-step minimum of the previous costs runs over all D
 *                       disparities (the reference leaves d = D-1 out).
 *   STM_SGM_TRUE_PATHS  every inverse diagonal pass starts at the true end of its forward
 *                       path, so every pixel receives each of the eight directions exactly
 *                       once (the reference's stored end points make some inverse passes
 *                       re-walk a corner diagonal and skip others). */
#define STM_SGM_FULL_MIN 8
#define STM_SGM_TRUE_PATHS 16
/* Non-finite inputs.  The aggregation's minima are v_min_f32 instructions, which ignore a NaN
 * operand; the reference's are strict-'<' scans (src/semiglobal.cpp:11-16,130), which keep a
 * NaN FIRST operand.  The two agree on every volume whose pixels all have a finite cost at
 * some d < D-1 (every volume stm_ssd_f32 / stm_birchfield_f32 produce: d = 0 is finite): no
 * NaN ever arises.  A pixel that is +inf at every d < D-1 makes the recurrence compute
 * (C + inf) - inf; from there on NaN and -inf spread along every path through it, and by
 * default WHICH of NaN / +-inf / a finite value lands on which element of those pixels'
 * sums is unspecified (pixels that no path reaches after such a pixel are unaffected).
 *   STM_SGM_EXACT_NONFINITE  reproduce the reference's placement of NaN and +-inf as well
 *                       (two more selects per element and step: 1-3 % slower). */
#define STM_SGM_EXACT_NONFINITE 32
/* Form selection.  By default the library picks per call between the paired form (a wave
 * per path; any shape) and the tiled form (row sweeps + 8x8-pixel tiles: the cost volume is
 * read four or five times, the sum never travels: 30-36 instead of 56-60 bytes per element;
 * covers D = 64, 128 or 256 without SINGLE_PASSES / EXACT_NONFINITE, chosen by default for
 * D >= 128 with enough columns in the batch to fill the chip).  Same results bit for bit.
 *   STM_SGM_PAIRED  never use the tiled form.
 *   STM_SGM_TILED   use the tiled form whenever it covers the call, whatever the batch size
 *                   (pointers must then be 16-byte aligned: STM_EINVAL otherwise; when the
 *                   library chooses the form itself, a call with unaligned pointers takes the
 *                   paired form instead).
 * Both flags must be passed identically to stm_sgm_workspace_bytes().
 *
 * Residency contract of the tiled form.  Its row sweeps exchange boundary values between the
 * workgroups of one launch, which must all be resident at once.  The library guarantees its
 * part: a sweep launch never has more workgroups than the device has CUs and the tiled form
 * is only ever selected when one sweep workgroup fits a CU (registers, LDS; the runtime's
 * occupancy calculator is asked).  The caller's part: while a tiled call runs, nothing else
 * holds CUs of the device for SECONDS -- no second tiled aggregation on another stream or in
 * another process, no persistent kernel.  Short kernels beside it (a copy, an RCCL
 * collective, other stereomatch ops) only delay it.  Every wait inside the sweeps is bounded
 * (~2 s); a wait that expires sets bit 0 of the STATUS WORD, the first 32-bit word of the
 * workspace (zeroed at the start of every call for which stm_sgm_uses_tiles() is 1), the wave
 * stops waiting, the call finishes and its results are INVALID.  The entry points never
 * synchronise, so the word is for the caller to read after its next synchronisation
 * (stm_sgm_status); RECOVERY: run the call again with STM_SGM_PAIRED, which needs nothing of
 * the sort.  The Python package does all of this (Semiglobal.status / check, Pipeline.status,
 * distributed.estimate_sharded re-runs a failed shard with the paired form).
 *   STM_SGM_COOPERATIVE  launch the sweeps with hipLaunchCooperativeKernel: the runtime
 *                   validates the grid against the kernel's occupancy and runs cooperative
 *                   launches one at a time per device, so tiled calls from several streams of
 *                   one process are safe beside each other.  Costs ~60 us per sweep launch and
 *                   the call's asynchrony (the runtime waits on the host for the stream's
 *                   earlier work before such a launch); ignored during stream capture. */
#define STM_SGM_PAIRED 64
#define STM_SGM_TILED 128
#define STM_SGM_COOPERATIVE 256
/*   STM_SGM_WALKS   use the tiled form with its border states from per-path WALKS instead of row
 *                   sweeps (one wave per path and direction reads the cost volume; seven or
 *                   eight reads instead of three, but no serial hand-off per image row and no
 *                   residency requirement: its status word is always 0).  What the library
 *                   picks by itself for single pairs and small batches at D >= 128;
 *                   STM_SGM_TILED means the sweep-fed variant wherever the device can hold it. */
#define STM_SGM_WALKS 512
/*   STM_SGM_SKEW    use the SKEWED-TILE form wherever it covers the call (D = 128 or 256, no ACCUMULATE /
 *                   SINGLE_PASSES / EXACT_NONFINITE): two passes over sheared 8x8 tiles, each
 *                   reading the cost volume once and carrying its four directions from tile to
 *                   tile (stm_sgm_skew.hip): ~20 instead of 30 bytes per element.  Bands of tiles
 *                   wait for the band above / below through progress counters; units are taken by
 *                   ticket, so there is NO residency requirement; the waits are bounded all the
 *                   same and report through the status word like the sweeps'.  What the library
 *                   picks by itself from two (D = 128) / four (D = 256) bands of tiles per
 *                   workgroup: 22 Teddy-sized pairs, eight 1080p pairs on an MI355X. */
#define STM_SGM_SKEW 1024
size_t stm_sgm_workspace_bytes(int H, int W, int D, int batch, int flags);
/* 1 when stm_sgm_f32 / stm_sgm_wta_f32 select the tiled form for this shape on the current
 * device, i.e. the workspace starts with the status word (a 256-byte header; also when such
 * a call then takes the paired form because of unaligned pointers), else 0. */
int stm_sgm_uses_tiles(int H, int W, int D, int batch, int flags);
/* Which form a call takes: STM_SGM_FORM_PAIRED, _SWEEPS (tiles behind row sweeps: the only one
 * whose status word can ever be set) or _WALKS (tiles behind per-path walks). */
#define STM_SGM_FORM_PAIRED 0
#define STM_SGM_FORM_SWEEPS 1
#define STM_SGM_FORM_WALKS 2
#define STM_SGM_FORM_SKEW 3
int stm_sgm_form(int H, int W, int D, int batch, int flags);
/* The status word's value given a HOST copy of the workspace's first 4 bytes (a helper for
 * bindings that cannot cast); 0 = ok. */
#define STM_SGM_STATUS_TIMEOUT 1u
unsigned stm_sgm_status(const void *host_copy_of_first_word);
int stm_sgm_f32(const float *cost, const float *left, float *out, int H, int W,
                int D, float penalty1, float penalty2, int flags,
                void *workspace, size_t workspace_bytes, int batch,
                void *stream);

/* ---- Fused aggregation + winner takes all (extension) -------------------------------
 * What Pipeline(SSD, WinnerTakesAll, Semiglobal).estimate computes (stereomatch/
 * pipeline.py:80-92 -> run_semiglobal then run_winners_take_all) in one call: the last
 * direction pair of the aggregation reduces every finished row to its first-minimum
 * disparity while it is still in registers, so the aggregated volume is neither written
 * nor re-read (saves 8 of ~72 bytes per (pixel, d) of the SSD+SGM+WTA pipeline).
 * `disparity` equals stm_sgm_f32 followed by stm_wta_f32 bit for bit.  `sum` is scratch
 * of the aggregated volume's size; on return it holds the sum of the first six
 * directions unless STM_SGM_KEEP_SUM asks for the finished sum.  Paired form only;
 * workspace as stm_sgm_workspace_bytes(..., flags).
 */
#define STM_SGM_KEEP_SUM 4
int stm_sgm_wta_f32(const float *cost, const float *left, float *sum,
                    int32_t *disparity, int H, int W, int D, float penalty1,
                    float penalty2, int flags, void *workspace,
                    size_t workspace_bytes, int batch, void *stream);

/* ---- Winner takes all (src/winners_take_all.cu:21-40) -------------------------
 * disparity[r][c] = index of the first minimum over d (strict '<' scan). */
int stm_wta_f32(const float *volume, int32_t *disparity, int H, int W, int D,
                int batch, void *stream);

/* ---- Dynamic programming (src/dynamic_programming.cu:24-68,166-181) ------------
 * path      [batch][H][W][D] int8  (scratch + output; fully written)
 * row_final [batch][H][D] float32  (output)
 * mode STM_DP_REFERENCE reproduces the CPU functor including its transposed
 * initial read (requires D <= H <= W, else STM_EINVAL) with the path volume
 * treated as zero-filled; STM_DP_CORRECTED initialises from column 0 of the
 * same row instead.
 */
#define STM_DP_REFERENCE 0
#define STM_DP_CORRECTED 1
int stm_dp_f32(const float *volume, int8_t *path, float *row_final,
               int32_t *disparity, int H, int W, int D, int mode, int batch,
               void *stream);

#ifdef __cplusplus
}
#endif
#endif /* STEREOMATCH_HIP_H_ */
