// Shared host/device helpers for libstereomatch_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "stereomatch_hip.h"

namespace stm {

// ---- host side: error reporting ---------------------------------------------
int fail(int code, const char *fmt, ...);
int check_launch(const char *what);
inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

#define STM_REQUIRE(cond, code, ...)            \
  do {                                          \
    if (!(cond)) return stm::fail(code, __VA_ARGS__); \
  } while (0)

constexpr int kWave = 64;

#if defined(__HIPCC__)
// ---- device side: wave64 cross-lane primitives on DPP ---------------------------
// dpp_ctrl encodings (gfx9): row_shr:n = 0x110+n, wave_shl:1 = 0x130,
// wave_shr:1 = 0x138, row_mirror = 0x140, row_half_mirror = 0x141,
// row_bcast:15 = 0x142, row_bcast:31 = 0x143, quad_perm = 0x00..0xff.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_f32(float old, float src) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                         __builtin_bit_cast(int, src), CTRL,
                                         ROW_MASK, BANK_MASK, false));
}
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ int dpp_i32(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, BANK_MASK, false);
}

// Value of lane-1 (lane 0 receives `fill`).
__device__ __forceinline__ float lane_up(float v, float fill) {
  return dpp_f32<0x138>(fill, v);  // wave_shr:1
}
// Value of lane+1 (lane 63 receives `fill`).
__device__ __forceinline__ float lane_down(float v, float fill) {
  return dpp_f32<0x130>(fill, v);  // wave_shl:1
}

__device__ __forceinline__ float read_lane(float v, int lane) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// v_min_f32 / v_min3_f32 as written: fminf() makes hipcc canonicalise both operands first
// (a v_max_f32 x, x, x each), which doubles the cost of min-heavy inner loops.  Same
// instruction, same result: the smaller operand, a NaN only when every operand is one.
__device__ __forceinline__ float vmin_f32(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmin3_f32(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// acc = value of lane-1 plus `addend` (a VGPR), written with one v_add_f32_dpp.  Lane 0 has
// no source and is skipped (bound_ctrl off), so it keeps what `acc` held: initialise `acc`
// to +inf once and reuse the same variable for every call, and lane 0 reads +inf for ever.
__device__ __forceinline__ void lane_up_add(float &acc, float v, float addend) {
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf"
      : "+v"(acc) : "v"(v), "v"(addend));
}
// acc = value of lane+1 plus `addend`; lane 63 keeps its (+inf) content.
__device__ __forceinline__ void lane_down_add(float &acc, float v, float addend) {
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf"
      : "+v"(acc) : "v"(v), "v"(addend));
}

// Minimum over the 64 lanes, returned wave-uniform.  NaNs are ignored (v_min_f32 IEEE
// semantics) unless every lane holds one; an all-+inf wave gives +inf.  Six v_min_f32_dpp in place: a lane whose DPP
// source does not exist (bound_ctrl off) is skipped and keeps its value.  The s_nop 1 in
// front of each covers the two wait states a DPP read needs after a VALU write of the same
// register (the assembler does not insert them inside inline asm).
__device__ __forceinline__ float wave_min_f32(float v) {
  asm("s_nop 1\n\t"
      "v_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"  // lane 15 of each row
      "s_nop 1\n\t"
      "v_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"  // rows 1, 3
      "s_nop 1\n\t"
      "v_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"        // rows 2, 3
      : "+v"(v));
  return read_lane(v, 63);
}

__device__ __forceinline__ int wave_min_i32(int v) {
  asm("s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
      : "+v"(v));
  return __builtin_amdgcn_readlane(v, 63);
}

// Minimum within aligned groups of G lanes (G = 16, 32 or 64), result valid in
// every lane of the group.  In-row steps use DPP mirrors, cross-row steps
// ds_bpermute through __shfl_xor.
template <int G>
__device__ __forceinline__ float group_min_f32(float v) {
  v = fminf(v, dpp_f32<0xb1>(v, v));   // quad_perm [1,0,3,2]
  v = fminf(v, dpp_f32<0x4e>(v, v));   // quad_perm [2,3,0,1]
  v = fminf(v, dpp_f32<0x141>(v, v));  // row_half_mirror
  v = fminf(v, dpp_f32<0x140>(v, v));  // row_mirror
  if (G >= 32) v = fminf(v, __shfl_xor(v, 16));
  if (G >= 64) v = fminf(v, __shfl_xor(v, 32));
  return v;
}
template <int G>
__device__ __forceinline__ int group_min_i32(int v) {
  int t;
  t = dpp_i32<0xb1>(v, v); v = t < v ? t : v;
  t = dpp_i32<0x4e>(v, v); v = t < v ? t : v;
  t = dpp_i32<0x141>(v, v); v = t < v ? t : v;
  t = dpp_i32<0x140>(v, v); v = t < v ? t : v;
  if (G >= 32) { t = __shfl_xor(v, 16); v = t < v ? t : v; }
  if (G >= 64) { t = __shfl_xor(v, 32); v = t < v ? t : v; }
  return v;
}

// Hand-off point of data that lanes of ONE wave pass to each other through LDS: everything the
// wave stored to LDS before this point is ordered before every LDS load after it (and loads
// before it stay before later stores).  LDS executes a wave's accesses in program order, so
// no instruction is needed; the wavefront-scope release/acquire pair plus the scheduling
// barrier stop the compiler from moving a may-not-alias load above the store it depends on.
__device__ __forceinline__ void wave_lds_handoff() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() is a release /
// acquire pair at workgroup scope over ALL address spaces: hipcc puts s_waitcnt vmcnt(0) in
// front of the s_barrier, which drains every global load and store the wave has in flight --
// in a kernel that keeps rows of prefetch outstanding across the barrier that is the whole
// memory latency, once per barrier.  Here only the LDS counter is drained; the "memory"
// clobber keeps the compiler from moving LDS (or any) accesses across.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ int wave_uniform(int v) {
  return __builtin_amdgcn_readfirstlane(v);
}

// ---- one pixel's disparities, VPL consecutive values per lane ---------------------
template <int VPL>
struct Row {
  float v[VPL];
};

// Loads the VPL disparities of one pixel owned by this lane; lanes/elements past
// D read as +inf so that they never win a minimum.
template <int VPL, bool VEC, bool NT = false>
__device__ __forceinline__ Row<VPL> load_row(const float *__restrict__ base, size_t pix,
                                              int D, int d0) {
  Row<VPL> r;
  const float *p = base + pix * (size_t)D + d0;
  if constexpr (VEC) {
    if constexpr (VPL % 4 == 0) {
#pragma unroll
      for (int j = 0; j < VPL; j += 4) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f q;
        if constexpr (NT) q = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p + j));
        else q = *reinterpret_cast<const v4f *>(p + j);
        r.v[j] = q.x; r.v[j + 1] = q.y; r.v[j + 2] = q.z; r.v[j + 3] = q.w;
      }
    } else if constexpr (VPL == 2) {
      typedef float v2f __attribute__((ext_vector_type(2)));
      v2f q;
      if constexpr (NT) q = __builtin_nontemporal_load(reinterpret_cast<const v2f *>(p));
      else q = *reinterpret_cast<const v2f *>(p);
      r.v[0] = q.x; r.v[1] = q.y;
    } else {
      r.v[0] = p[0];
    }
  } else {
#pragma unroll
    for (int j = 0; j < VPL; ++j) r.v[j] = (d0 + j < D) ? p[j] : __builtin_inff();
  }
  return r;
}

// The same with the row given as an element offset (a running 64-bit scalar in the callers).
template <int VPL, bool VEC, bool NT = false>
__device__ __forceinline__ Row<VPL> load_row_at(const float *__restrict__ base, long off, int D,
                                                 int d0) {
  return load_row<VPL, VEC, NT>(base + off, 0, D, d0);
}

template <int VPL, bool VEC, bool NT = false>
__device__ __forceinline__ void store_row(float *__restrict__ base, size_t pix, int D, int d0,
                                          const Row<VPL> &r) {
  float *p = base + pix * (size_t)D + d0;
  if constexpr (VEC) {
    if constexpr (VPL % 4 == 0) {
#pragma unroll
      for (int j = 0; j < VPL; j += 4) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f q; q.x = r.v[j]; q.y = r.v[j + 1]; q.z = r.v[j + 2]; q.w = r.v[j + 3];
        if constexpr (NT) __builtin_nontemporal_store(q, reinterpret_cast<v4f *>(p + j));
        else *reinterpret_cast<v4f *>(p + j) = q;
      }
    } else if constexpr (VPL == 2) {
      typedef float v2f __attribute__((ext_vector_type(2)));
      v2f q; q.x = r.v[0]; q.y = r.v[1];
      if constexpr (NT) __builtin_nontemporal_store(q, reinterpret_cast<v2f *>(p));
      else *reinterpret_cast<v2f *>(p) = q;
    } else {
      p[0] = r.v[0];
    }
  } else {
#pragma unroll
    for (int j = 0; j < VPL; ++j)
      if (d0 + j < D) p[j] = r.v[j];
  }
}

template <int VPL, bool VEC, bool NT = false>
__device__ __forceinline__ void store_row_at(float *__restrict__ base, long off, int D, int d0,
                                             const Row<VPL> &r) {
  store_row<VPL, VEC, NT>(base + off, 0, D, d0, r);
}

// ---- buffer addressing -------------------------------------------------------------
// A wave-uniform 128-bit descriptor (base, byte size) in SGPRs, a per-lane 32-bit byte offset
// and a wave-uniform 32-bit byte offset: a row address costs one scalar add and no vector
// arithmetic (a 64-bit global address costs a vector 64-bit add per load), small constant
// parts fold into the instruction.  Regions must be smaller than 4 GiB.
typedef __amdgpu_buffer_rsrc_t BufRsrc;
__device__ __forceinline__ BufRsrc make_rsrc(const void *base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
template <int VPL>
__device__ __forceinline__ Row<VPL> buf_load_row(BufRsrc r, int voff, int soff) {
  Row<VPL> o;
  if constexpr (VPL == 4) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u q = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    // (Elements go through scalars: __builtin_bit_cast applied to a vector ELEMENT expression,
    // `__builtin_bit_cast(float, q.y)`, reads element 0 with this hipcc -- every lane then holds
    // x, x, x, x and the load is narrowed to one dword.)
    const unsigned q0 = q.x, q1 = q.y, q2 = q.z, q3 = q.w;
    o.v[0] = __builtin_bit_cast(float, q0); o.v[1] = __builtin_bit_cast(float, q1);
    o.v[2] = __builtin_bit_cast(float, q2); o.v[3] = __builtin_bit_cast(float, q3);
  } else if constexpr (VPL == 2) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const v2u q = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    const unsigned q0 = q.x, q1 = q.y;
    o.v[0] = __builtin_bit_cast(float, q0); o.v[1] = __builtin_bit_cast(float, q1);
  } else {
    static_assert(VPL == 1, "whole 4-, 8- or 16-byte rows per lane");
    o.v[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  }
  return o;
}
// A buffer store reads its data registers for some cycles after it has issued, and a VALU write to
// them in the very next slots races with that read.  hipcc's hazard recognizer pads the case
// only when the store's soffset is an immediate: with an SGPR there it assumes the issue takes
// long enough, which does not hold on gfx950 -- a 16-byte store followed at once by the next
// row's arithmetic in the same registers wrote that arithmetic's values to memory for the last
// lanes (found as mailbox rows whose tags matched and whose values did not, and as column-line
// states holding intensity differences).  After every buffer store with a register soffset: four
// wait states during which the data registers are still in use as far as the compiler knows.
template <int VPL>
__device__ __forceinline__ void hold_store_data(const Row<VPL> &x) {
  if constexpr (VPL == 4) asm volatile("s_nop 3" ::"v"(x.v[0]), "v"(x.v[1]), "v"(x.v[2]), "v"(x.v[3]) : "memory");
  else if constexpr (VPL == 2) asm volatile("s_nop 3" ::"v"(x.v[0]), "v"(x.v[1]) : "memory");
  else asm volatile("s_nop 3" ::"v"(x.v[0]) : "memory");
}
template <int VPL>
__device__ __forceinline__ void buf_store_row(BufRsrc r, int voff, int soff, const Row<VPL> &x) {
  if constexpr (VPL == 4) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    v4u q;
    q.x = __builtin_bit_cast(unsigned, x.v[0]); q.y = __builtin_bit_cast(unsigned, x.v[1]);
    q.z = __builtin_bit_cast(unsigned, x.v[2]); q.w = __builtin_bit_cast(unsigned, x.v[3]);
    __builtin_amdgcn_raw_buffer_store_b128(q, r, voff, soff, 0);
  } else if constexpr (VPL == 2) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    v2u q;
    q.x = __builtin_bit_cast(unsigned, x.v[0]); q.y = __builtin_bit_cast(unsigned, x.v[1]);
    __builtin_amdgcn_raw_buffer_store_b64(q, r, voff, soff, 0);
  } else {
    static_assert(VPL == 1, "whole 4-, 8- or 16-byte rows per lane");
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x.v[0]), r, voff, soff, 0);
  }
  hold_store_data<VPL>(x);
}

// 16-byte rows through a descriptor with a cache policy (aux bits of the raw buffer builtins:
// kAuxSc1 = sc1, i.e. served by / written through to the memory side: data that another
// workgroup of the SAME launch wrote or will read -- MI355X guide, Guideline 16).  An offset at or
// beyond the descriptor's size (kNowhere) is an access the hardware drops: loads return 0.
constexpr int kBufSc1 = 16;
constexpr unsigned kNowhere = 0xfffffff0u;
template <int AUX>
__device__ __forceinline__ Row<4> buf_load_row4(BufRsrc r, unsigned voff, unsigned soff) {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  const v4u q = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, AUX);
  const unsigned q0 = q.x, q1 = q.y, q2 = q.z, q3 = q.w;  // (see buf_load_row)
  Row<4> o;
  o.v[0] = __builtin_bit_cast(float, q0); o.v[1] = __builtin_bit_cast(float, q1);
  o.v[2] = __builtin_bit_cast(float, q2); o.v[3] = __builtin_bit_cast(float, q3);
  return o;
}
template <int AUX>
__device__ __forceinline__ void buf_store_row4(BufRsrc r, unsigned voff, unsigned soff, const Row<4> &x) {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  v4u q;
  q.x = __builtin_bit_cast(unsigned, x.v[0]); q.y = __builtin_bit_cast(unsigned, x.v[1]);
  q.z = __builtin_bit_cast(unsigned, x.v[2]); q.w = __builtin_bit_cast(unsigned, x.v[3]);
  __builtin_amdgcn_raw_buffer_store_b128(q, r, (int)voff, (int)soff, AUX);
  hold_store_data<4>(x);
}

// Loads / stores of data that another workgroup of the SAME launch has written / will read:
// relaxed atomics at agent scope, dword by dword (sc1 accesses: served by / written through to
// the memory side.  MI355X guide, Guideline 16: the per-XCD L2s are not coherent with each
// other).  For hand-overs that happen once per thousands of pixels; ordering against the flag
// that announces them is the caller's (s_waitcnt + release on one side, acquire on the other).
template <int VPL>
__device__ __forceinline__ Row<VPL> load_row_agent(const float *p) {
  Row<VPL> o;
#pragma unroll
  for (int j = 0; j < VPL; ++j)
    o.v[j] = __hip_atomic_load(const_cast<float *>(p) + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return o;
}
template <int VPL>
__device__ __forceinline__ void store_row_agent(float *p, const Row<VPL> &x) {
#pragma unroll
  for (int j = 0; j < VPL; ++j) __hip_atomic_store(p + j, x.v[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#endif  // __HIPCC__

}  // namespace stm
