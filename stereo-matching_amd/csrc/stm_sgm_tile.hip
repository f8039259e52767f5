// Semiglobal aggregation, tiled form: the tile kernel for D = 64 and D = 256, one line of a
// direction pair per wave (see stm_sgm_tiled.hip for the whole form, stm_sgm_tile_common.h for
// the chains and the update).
#include "stm_sgm_tile_common.h"

namespace stm {
namespace {

// The sequence of tiles a workgroup processes.  Static mode: tiles blockIdx.x, + gridDim.x, ...
// (row-major over tiles, images one after the other).  Segment mode (P.by_rows: the call's
// border states come from row sweeps and no walk of the left-to-right direction has run): units
// are segments of tile rows (stm_sgm_tiled_index.h, row_segs), taken by an atomic ticket, walked
// left to right.
struct TilePos {
  int b, trow, tcol;
  int tbeg, tend, seg;  // segment mode: the unit's tile columns [tbeg, tend) and its number in the row
};

// Waits for a flag another workgroup of this launch sets (bounded, as the sweeps' waits: ~2 s,
// then the status word is set and nobody waits again).
__device__ __forceinline__ void wait_flag(const unsigned *flag, unsigned *status) {
  if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    __builtin_amdgcn_s_sleep(4);
    if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
      if ((threadIdx.x & 63) == 0) atomicOr(status, STM_SGM_STATUS_TIMEOUT);
      return;
    }
  }
}

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
  __shared__ int s_unit;  // segment mode: the ticket of the workgroup's next unit
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
  const bool rows = P.by_rows != 0;
  const RowSegs RS = row_segs(G, P.segs);
  const int nunits = seg_units(G, RS);
  unsigned *status = reinterpret_cast<unsigned *>(ws + P.status_off);
  unsigned *flags = reinterpret_cast<unsigned *>(ws + G.flag_off);
  TileWalk sw;  // static mode
  int id = (int)blockIdx.x;
  TilePos cur_t, next_t;  // the tile being computed / the one being prefetched
  auto unit_first = [&](int unit, TilePos &t) {
    const SegUnit u = seg_unit(G, RS, unit);
    t.b = u.b; t.trow = u.trow; t.tcol = u.tbeg; t.tbeg = u.tbeg; t.tend = u.tend; t.seg = u.seg;
  };
  bool valid;
  if (rows) {
    // Every unit, the first included, is taken by ticket: a unit is only ever held by a workgroup
    // that runs, so a segment never waits for one that has not started (wait_flag).
    if (threadIdx.x == 0) s_unit = (int)atomicAdd(reinterpret_cast<unsigned *>(ws + P.status_off) + kTicketWord, 1u);
    __syncthreads();
    id = s_unit;
    __syncthreads();
    valid = id < nunits;
    unit_first(valid ? id : 0, cur_t);
  } else {
    valid = id < ntiles;
    sw.init(valid ? id : 0, stride, G);
    cur_t.b = sw.b; cur_t.trow = sw.trow; cur_t.tcol = sw.tcol;
    cur_t.tbeg = 0; cur_t.tend = 0; cur_t.seg = 0;
  }
  if (!valid) return;
  next_t = cur_t;
  // The tile after cur_t (false: there is none, next_t stays on cur_t and the prefetches re-read
  // it).  `next_unit`: the ticket fetched at the top of a unit's last tile.
  auto advance = [&](int next_unit) {
    if (!rows) {
      if (id + stride >= ntiles) return false;
      sw.next(G);
      next_t.b = sw.b; next_t.trow = sw.trow; next_t.tcol = sw.tcol;
      return true;
    }
    if (cur_t.tcol + 1 < cur_t.tend) {
      next_t = cur_t;
      ++next_t.tcol;
      return true;
    }
    if (next_unit >= nunits) return false;
    unit_first(next_unit, next_t);
    return true;
  };

  if (grp == 0) {
    Row<VPL> cpre[kT];   // next tile: tile row w of the cost tile
    float ipre = 0.0f;   // next tile: this thread's element of the intensity patch
    Entries<VPL, 0> E0;  // next tile: border states of pair 0
    Row<VPL> carry;      // segment mode: the rightward state with which row w left the previous tile
    bool have_carry = false;
    // Every prefetch redefines all of its registers unconditionally (past the last tile it
    // re-reads that tile, rows and samples outside a partial tile are clamped into it): a
    // conditional load would keep the previous contents alive around the whole loop.
    auto prefetch = [&](const TilePos &t) {
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
    auto stage = [&](const TilePos &t) {
      const int th = imin(kT, G.H - t.trow * kT);
      if (w < th) {
#pragma unroll
        for (int c = 0; c < kT; ++c) store_row<VPL, true>(Ct + (size_t)(w * kT + c) * D, 0, D, d0, cpre[c]);
      }
      if (threadIdx.x < (kT + 2) * (kT + 2)) Ip[threadIdx.x] = ipre;
    };
    prefetch(next_t);
    stage(next_t);
    for (;;) {
      const int b = cur_t.b, x0 = cur_t.tcol * kT, y0 = cur_t.trow * kT;
      const int tw = imin(kT, G.W - x0), th = imin(kT, G.H - y0);
      const bool full = tw == kT && th == kT;
      float *out = out_all ? out_all + (size_t)b * npix * D : nullptr;
      int32_t *disp = disp_all ? disp_all + (size_t)b * npix : nullptr;
      STM_STAMP_TILE_ID((cur_t.b * G.nTR + cur_t.trow) * G.nTC + cur_t.tcol);
      STM_STAMP(0);
      Entries<VPL, 0> E0now = E0;
      const bool last_of_unit = rows && cur_t.tcol + 1 == cur_t.tend;
      if (rows) {
        // The rightward entry state of row w: carried from the previous tile of the segment; at
        // the start of a segment that is not the row's first, what the segment on the left stored
        // (its flag says when; sc1: another workgroup's store, possibly on another XCD).
        if (have_carry) {
          E0now.e[0] = carry;
        } else if (cur_t.tcol > 0) {
          wait_flag(flags + seg_flag_index(G, b, cur_t.trow, cur_t.seg - 1), status);
          if (w < th) {
            const float *src = colb_ptr(ws, G, b, 0, cur_t.tcol - 1, y0 + w);
            E0now.e[0] = load_row_agent<VPL>(src + d0);
          }
        }
        if (last_of_unit && w16 == 0 && lane == 0)
          s_unit = (int)atomicAdd(reinterpret_cast<unsigned *>(ws + P.status_off) + kTicketWord, 1u);
      }
      // (the previous tile's winner-takes-all has read the sum tile: both groups are past B4)
      if (P.accumulate) stage_accumulate_row<VPL>(St, out, G, x0, y0, tw, th, w, d0);
      STM_STAMP(1);
      lds_barrier();  // B0
      STM_STAMP(2);
      const int next_unit = last_of_unit ? s_unit : 0;
      // (pair 2's entry states are requested before pair 0: a load from HBM takes as long as a pair)
      const PhaseGeom g2 = tile_geom(2, w, tw, th);
      Entries<VPL, 2> E2;
      load_entries<VPL, 2>(E2, g2, border_base(ws, G, b), G, x0, y0, d0);
      tile_pair<VPL, 0, FINAL>(tile_geom(0, w, tw, th), E0now, full, Ct, St, Ip, P, x0, y0, out, disp, X, lane,
                               [] {}, &carry);  // (unconditionally: a pointer that may be null keeps `carry` in scratch memory)
      have_carry = rows && tw == kT && !last_of_unit;
      // A segment's last tile leaves its rows' states for the segment on its right.
      const bool hand_over = last_of_unit && cur_t.tend < G.nTC;
      if (hand_over) {
        if (w < th) {
          float *dst = colb_ptr(ws, G, b, 0, cur_t.tcol, y0 + w);
          store_row_agent<VPL>(dst + d0, carry);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (before B1, behind which the flag is set)
      }
      STM_STAMP(3);
      lds_barrier();  // B1
      if (hand_over && w16 == 0 && lane == 0)
        __hip_atomic_store(flags + seg_flag_index(G, b, cur_t.trow, cur_t.seg), 1u, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_AGENT);
      STM_STAMP(4);
      tile_pair<VPL, 2, FINAL>(g2, E2, full, Ct, St, Ip, P, x0, y0, out, disp, X, lane, [&] {
        STM_STAMP(5);
        lds_barrier();  // B2
      });
      STM_STAMP(6);
      const bool more = advance(next_unit);
      prefetch(next_t);
      lds_barrier();  // B3
      stage(next_t);  // (while group 1 runs its last update)
      if constexpr (FINAL != 0) {
        lds_barrier();  // B4: the finished sum tile is complete
        tile_wta_lds<D, 2 * kT>(St, w16, lane, tw, th, G, x0, y0, disp);
      } else if (P.accumulate) {
        lds_barrier();  // B4: group 1's last update has read the sum tile the next tile's staging overwrites
      }
      STM_STAMP(7);
      if (!more) break;
      cur_t = next_t;
      id += stride;
    }
  } else {
    Entries<VPL, 1> E1;  // next tile: border states of pair 1
    auto prefetch = [&](const TilePos &t) {
      const int x0 = t.tcol * kT, y0 = t.trow * kT;
      const int tw = imin(kT, G.W - x0), th = imin(kT, G.H - y0);
      load_entries<VPL, 1>(E1, tile_geom(1, w, tw, th), border_base(ws, G, t.b), G, x0, y0, d0);
    };
    prefetch(next_t);
    for (;;) {
      const int b = cur_t.b, x0 = cur_t.tcol * kT, y0 = cur_t.trow * kT;
      const int tw = imin(kT, G.W - x0), th = imin(kT, G.H - y0);
      const bool full = tw == kT && th == kT;
      float *out = out_all ? out_all + (size_t)b * npix * D : nullptr;
      int32_t *disp = disp_all ? disp_all + (size_t)b * npix : nullptr;
      STM_STAMP_TILE_ID((cur_t.b * G.nTR + cur_t.trow) * G.nTC + cur_t.tcol);
      STM_STAMP(0);
      const Entries<VPL, 1> E1now = E1;
      const bool last_of_unit = rows && cur_t.tcol + 1 == cur_t.tend;
      STM_STAMP(1);
      lds_barrier();  // B0
      STM_STAMP(2);
      const int next_unit = last_of_unit ? s_unit : 0;
      const PhaseGeom g3 = tile_geom(3, w, tw, th);
      Entries<VPL, 3> E3;
      load_entries<VPL, 3>(E3, g3, border_base(ws, G, b), G, x0, y0, d0);
      tile_pair<VPL, 1, FINAL>(tile_geom(1, w, tw, th), E1now, full, Ct, St, Ip, P, x0, y0, out, disp, X, lane,
                               [&] {
                                 STM_STAMP(3);
                                 lds_barrier();  // B1
                               });
      STM_STAMP(4);
      lds_barrier();  // B2
      const bool more = advance(next_unit);
      tile_pair<VPL, 3, FINAL>(g3, E3, full, Ct, St, Ip, P, x0, y0, out, disp, X, lane, [&] {
        prefetch(next_t);
        STM_STAMP(5);
        lds_barrier();  // B3
        STM_STAMP(6);
      });
      if constexpr (FINAL != 0) {
        lds_barrier();  // B4
        tile_wta_lds<D, 2 * kT>(St, w16, lane, tw, th, G, x0, y0, disp);
      } else if (P.accumulate) {
        lds_barrier();  // B4
      }
      STM_STAMP(7);
      if (!more) break;
      cur_t = next_t;
      id += stride;
    }
  }
}

}  // namespace

int tiled_launch_tiles(const float *cost, const float *left, float *out, int32_t *disp, float *ws,
                       const TiledCall &c, int final_mode, hipStream_t s) {
  const TiledGeom &G = c.G;
  TileArgs A;
  A.G = G; A.p1 = c.p1; A.p2 = c.p2; A.dlim = c.dlim; A.true_paths = c.true_paths; A.accumulate = c.accumulate;
  A.stamps = c.tile_stamps;
  A.by_rows = c.by_rows ? 1 : 0;
  A.segs = c.segs;
  A.status_off = c.status_off;
  // Persistent workgroups; tiles are dealt round robin (segments of tile rows by ticket behind
  // sweeps).  D = 256: one sixteen-wave workgroup per CU (the LDS allows no more); D = 64: two.
  const long long units = c.by_rows ? (long long)seg_units(G, row_segs(G, c.segs)) : (long long)G.nTR * G.nTC * G.batch;
  const long long wgs = (long long)c.cus * (G.D == 256 ? 1 : 2);
  const dim3 grid((unsigned)(units < wgs ? units : wgs)), block(2 * kT * 64);
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
