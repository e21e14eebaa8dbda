// build: hipcc -O3 -w --offload-arch=gfx950 tools/hipbench/lds_atomic_rate.hip -o /tmp/lds_atomic_rate
// LDS atomic throughput on gfx950: lane-operations per clock per CU for ds_add_f32 / ds_add_u32 / ds_add_u64,
// conflict-free addressing (lane -> own bank), 16 waves per workgroup, one workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
  __shared__ unsigned long long tile[8192];
  float* tf = reinterpret_cast<float*>(tile);
  unsigned* tu = reinterpret_cast<unsigned*>(tile);
  for (int i = threadIdx.x; i < 8192; i += 1024) tile[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int it = 0; it < iters; ++it) {
    const int cell = (wave * 7 + it * 13) & 127;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = ((cell + u * 16) & 127) * 64 + lane;
      if (MODE == 0) atomicAdd(&tf[idx], 1.0f);
      if (MODE == 1) atomicAdd(&tu[idx], 1u);
      if (MODE == 2) atomicAdd(&tile[idx], 1ull);
      if (MODE == 3) tf[idx] += 1.0f;      // plain read-modify-write (racy, rate reference)
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = tf[5] + (float)tu[7];
}
template <int MODE>
void run(const char* name) {
  float* out; hipMalloc(&out, 4096);
  const int iters = 2000, blocks = 256;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, out, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double lane_ops = (double)iters * 8 * 1024;           // per workgroup (= per CU)
  printf("%-12s %.3f ms  %.2f lane-ops/clk/CU (at 2.4 GHz)\n", name, ms, lane_ops / (ms * 1e-3) / 2.4e9);
}
int main() { run<0>("ds_add_f32"); run<1>("ds_add_u32"); run<2>("ds_add_u64"); run<3>("plain rmw"); return 0; }
