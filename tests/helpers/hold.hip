// Test helper (not part of the product library): a kernel that holds CUs for a given time, so
// that tests can launch the tiled aggregation beside "somebody else's long-running kernel".
// Every workgroup allocates 96 KB of LDS -- no second workgroup of anything LDS-heavy fits next
// to it on its CU (the row sweeps need 64 KB) -- and spins on s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(64) void hold_kernel(unsigned long long ticks, unsigned *sink) {
  __shared__ unsigned pad[24 * 1024];
  pad[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (pad[threadIdx.x] == 0xffffffffu) *sink = 1;  // never true: keeps `pad` allocated
}

extern "C" int stm_test_hold_cus(int workgroups, int milliseconds, void *stream) {
  if (workgroups <= 0 || milliseconds <= 0 || milliseconds > 20000) return 1;
  hipLaunchKernelGGL(hold_kernel, dim3(workgroups), dim3(64), 0, static_cast<hipStream_t>(stream),
                     (unsigned long long)milliseconds * 100000ull, (unsigned *)nullptr);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
