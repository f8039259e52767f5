// Stand-alone check of the cross-lane primitives of the two-pixels-per-wave step (stm_sgm_step.h):
// butterfly minimum over 32 lanes with v_permlane16_swap, row_newbcast, neighbour shifts inside a
// half.  tools/build_probe.sh microbench builds it as tools/_probe/h2_dpp_check.
#include <hip/hip_runtime.h>
template <int S>
__device__ __forceinline__ float nb(float pen, float m) {
  float r;
  asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(pen), "v"(m), "n"(S));
  return r;
}
__device__ __forceinline__ float half_min_f32(float v) {
  float t;
  asm("s_nop 1\n\t"
      "v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_mov_b32 %1, %0\n\t"
      "s_nop 1\n\t"
      "v_permlane16_swap_b32 %1, %0\n\t"
      "s_nop 1\n\t"
      "v_min_f32 %0, %0, %1"
      : "+v"(v), "=&v"(t));
  return v;
}
__global__ void k(float* o, const float* in) {
  float v = in[threadIdx.x];
  float m = half_min_f32(v);
  float f = nb<3>(v, m);
  float g = nb<11>(v, 0.0f);
  float below = __builtin_inff(), above = __builtin_inff();
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %2, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %3, %4 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %2, %4 wave_shr:1 row_mask:0xa bank_mask:0x1\n\t"
      "v_add_f32_dpp %1, %3, %4 wave_shl:1 row_mask:0x5 bank_mask:0x8"
      : "+v"(below), "+v"(above) : "v"(v), "v"(v), "v"(0.0f));
  o[threadIdx.x] = m; o[64 + threadIdx.x] = f; o[128 + threadIdx.x] = g; o[192 + threadIdx.x] = below; o[256 + threadIdx.x] = above;
}
int main() {
  float h[64], r[320]; for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37) % 64) + 100.0f;
  float *di, *dout; hipMalloc(&di, 256); hipMalloc(&dout, 1280); hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dout, di); hipMemcpy(r, dout, 1280, hipMemcpyDeviceToHost);
  int bad = 0;
  float m0 = 1e9, m1 = 1e9; for (int i = 0; i < 32; ++i) { m0 = fminf(m0, h[i]); m1 = fminf(m1, h[32 + i]); }
  for (int i = 0; i < 64; ++i) {
    float m = i < 32 ? m0 : m1;
    if (r[i] != m) { printf("min lane %d got %f want %f\n", i, r[i], m); bad++; }
    float pen3 = h[(i & ~15) + 3], pen11 = h[(i & ~15) + 11];
    if (r[64 + i] != pen3 + m) { printf("nb3 lane %d got %f want %f\n", i, r[64 + i], pen3 + m); bad++; }
    if (r[128 + i] != pen11) { printf("nb11 lane %d got %f want %f\n", i, r[128 + i], pen11); bad++; }
    float wb = (i & 31) == 0 ? INFINITY : h[i - 1], wa = (i & 31) == 31 ? INFINITY : h[i + 1];
    if (r[192 + i] != wb) { printf("below lane %d got %f want %f\n", i, r[192 + i], wb); bad++; }
    if (r[256 + i] != wa) { printf("above lane %d got %f want %f\n", i, r[256 + i], wa); bad++; }
  }
  printf("bad=%d\n", bad);
  return bad != 0;
}
