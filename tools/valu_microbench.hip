#include <hip/hip_runtime.h>
#include <stdio.h>
// Microbenchmark: what a wave64 VALU instruction costs on gfx950, by kind and by waves per SIMD
// (DESIGN.md section 4.1b cites its output).  Each kernel runs N iterations of a block of 64
// instructions.  Build and run: tools/build_probe.sh microbench, then tools/_probe/valu_microbench
// on the GPU box.
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int MODE>
__global__ void k(float* o, int iters) {
  float a = threadIdx.x, b = 1.0f, c = 2.0f, d = 3.0f;
  float2 p = {a, b}, q = {c, d};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { asm volatile(REP64("v_add_f32 %0, %0, %1\n\t") : "+v"(a) : "v"(b)); }            // dependent chain
    if (MODE == 1) { asm volatile(REP8(REP8("v_add_f32 %0, %0, %4\n\t") ) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(1.0f)); }
    if (MODE == 2) { asm volatile(REP8("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4\n\tv_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4\n\t") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(1.0f)); }  // 4 independent chains
    if (MODE == 3) { asm volatile(REP64("v_pk_add_f32 %0, %0, %1\n\t") : "+v"(p) : "v"(q)); }          // dependent pk
    if (MODE == 4) { asm volatile(REP64("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t") : "+v"(a)); }  // dependent dpp with nops
    if (MODE == 5) { asm volatile(REP64("v_min3_f32 %0, %0, %1, %2\n\t") : "+v"(a) : "v"(b), "v"(c)); }
    if (MODE == 6) { asm volatile(REP8("v_add_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf\n\t") : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }  // independent dpp adds
    if (MODE == 7) { asm volatile(REP8("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %1, %0\n\ts_nop 1\n\tv_min_f32 %0, %0, %1\n\tv_add_f32 %2, %2, %3\n\tv_add_f32 %2, %2, %3\n\tv_add_f32 %2, %2, %3\n\tv_add_f32 %2, %2, %3\n\tv_add_f32 %2, %2, %3\n\t") : "+v"(a), "+v"(b), "+v"(c) : "v"(d)); } // 8 instr
  }
  o[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + p.x + p.y;
}
template <int MODE>
void run(const char* name, int per_iter) {
  float* o; hipMalloc(&o, 256 * 1024 * 4 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int wps = 1; wps <= 8; wps *= 2) {   // waves per SIMD
    dim3 grid(256), block(64 * 4 * wps > 1024 ? 1024 : 64 * 4 * wps);
    int blocks_per_cu = (64 * 4 * wps + 1023) / 1024;
    grid.x = 256 * blocks_per_cu;
    k<MODE><<<grid, block>>>(o, 10);
    hipEventRecord(e0);
    k<MODE><<<grid, block>>>(o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ns_per_instr_per_wave = ms * 1e6 / ((double)iters * per_iter);
    printf("%-28s waves/SIMD %d: %.2f ns per instr per wave (%.2f ns per instr per SIMD)\n", name, wps, ns_per_instr_per_wave, ns_per_instr_per_wave / wps);
  }
  hipFree(o);
}
int main() {
  run<0>("dependent v_add", 64);
  run<1>("dependent v_add (b)", 64);
  run<2>("4 independent v_add chains", 64);
  run<3>("dependent v_pk_add", 64);
  run<4>("dependent dpp min + s_nop 1", 64);
  run<5>("dependent v_min3", 64);
  run<6>("independent dpp adds", 64);
  run<7>("mov+swap+min + 5 adds (8 v)", 64);
  return 0;
}
