// Access-pattern probe: what bandwidth do wave-per-path walks reach without any compute?
// hipcc --offload-arch=gfx950 -O3 -o walk_probe walk_probe.hip && ./walk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int H = 1080, W = 1920, D = 256, B = 2;

// mode 0: horizontal paths (wave = row), 1: vertical paths (wave = column)
// PF pixels in flight per wave; each step loads 1 KB (float4 per lane) and accumulates.
template <int PF, bool STORE>
__global__ __launch_bounds__(256) void walk(const float4 *__restrict__ C, float4 *__restrict__ S,
                                            float *__restrict__ sink, int mode) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int paths = mode == 0 ? H : W;
  if (wave >= paths * B) return;
  const int b = wave / paths, w = wave % paths;
  const long len = mode == 0 ? W : H;
  const long step = mode == 0 ? 1 : W;                 // in pixels
  long pix = (long)b * H * W + (mode == 0 ? (long)w * W : w);
  float4 buf[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) buf[u] = C[(pix + u * step) * (D / 4) + lane];
  float4 acc = make_float4(0, 0, 0, 0);
  for (long i = 0; i < len; i += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (i + u >= len) break;   // wave-uniform
      const float4 v = buf[u];
      const long nxt = i + u + PF < len ? i + u + PF : len - 1;
      buf[u] = C[(pix + (nxt - i) * step) * (D / 4) + lane];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      if (STORE) S[(pix + u * step) * (D / 4) + lane] = acc;
    }
    pix += PF * step;
  }
  if (acc.x == 123.456f) sink[0] = acc.y;
}

// Plain streaming reference: grid-stride float4 copy / read.
__global__ void stream_copy(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

int main() {
  const size_t n4 = (size_t)B * H * W * D / 4;
  float4 *C, *S; float *sink;
  CK(hipMalloc(&C, n4 * 16)); CK(hipMalloc(&S, n4 * 16)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(C, 0, n4 * 16)); CK(hipMemset(S, 0, n4 * 16));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double gb = n4 * 16 / 1e9;
  auto timeit = [&](const char *name, double bytes_gb, auto launch) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-44s %7.3f ms  %6.2f TB/s\n", name, ms, bytes_gb / ms);
  };
  timeit("stream copy (read+write)", 2 * gb, [&] { hipLaunchKernelGGL(stream_copy, dim3(4096), dim3(256), 0, 0, C, S, n4); });
  for (int mode = 0; mode < 2; ++mode) {
    const int paths = (mode == 0 ? H : W) * B;
    const dim3 grid((paths + 3) / 4);
    char name[96];
    snprintf(name, 96, "%s walk, read only, PF=4", mode ? "vertical  " : "horizontal");
    timeit(name, gb, [&] { hipLaunchKernelGGL((walk<4, false>), grid, dim3(256), 0, 0, C, S, sink, mode); });
    snprintf(name, 96, "%s walk, read only, PF=16", mode ? "vertical  " : "horizontal");
    timeit(name, gb, [&] { hipLaunchKernelGGL((walk<16, false>), grid, dim3(256), 0, 0, C, S, sink, mode); });
    snprintf(name, 96, "%s walk, read+write, PF=4", mode ? "vertical  " : "horizontal");
    timeit(name, 2 * gb, [&] { hipLaunchKernelGGL((walk<4, true>), grid, dim3(256), 0, 0, C, S, sink, mode); });
    snprintf(name, 96, "%s walk, read+write, PF=16", mode ? "vertical  " : "horizontal");
    timeit(name, 2 * gb, [&] { hipLaunchKernelGGL((walk<16, true>), grid, dim3(256), 0, 0, C, S, sink, mode); });
  }
  return 0;
}
