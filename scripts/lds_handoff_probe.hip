// Micro-benchmark: latency of a wave-to-wave hand-off inside one workgroup through LDS, and between workgroups through
// global memory (agent-scope relaxed atomics), the two hops the triangular sweeps of schwarz.hpp are made of.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_handoff_probe lds_handoff_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) unsigned long long gu64_t;
constexpr unsigned long long kS = 0xFFF4A5A5DEADBEEFull;

// one workgroup of W waves; token t is produced by wave t % W after it has seen token t-1 in LDS slot (t-1) % 64
template <int EXTRA>
__global__ __launch_bounds__(1024) void k_lds_chain(int hops, long long *out, unsigned long long *g) {
  __shared__ unsigned long long s[64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, W = blockDim.x >> 6;
  if (threadIdx.x < 64) s[threadIdx.x] = kS;
  __syncthreads();
  const long long t0 = wall_clock64();
  for (int t = wave; t < hops; t += W) {
    if (t > 0) {
      while (__hip_atomic_load(&s[(t - 1) & 63], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != (unsigned long long)(t - 1)) {
        if (EXTRA & 1) __builtin_amdgcn_s_sleep(1);
      }
    }
    if (lane == 0) {
      __hip_atomic_store(&s[t & 63], (unsigned long long)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (EXTRA & 2) __hip_atomic_store((gu64_t *)(g + t), (unsigned long long)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[0] = wall_clock64() - t0;
}

// B workgroups of one wave; token t is produced by block t % B after it has seen token t-1 in global memory
__global__ __launch_bounds__(64) void k_glb_chain(int hops, long long *out, unsigned long long *g) {
  const int B = gridDim.x;
  const long long t0 = wall_clock64();
  for (int t = blockIdx.x; t < hops; t += B) {
    if (t > 0) {
      while (__hip_atomic_load((gu64_t *)(g + t - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)(t - 1)) __builtin_amdgcn_s_sleep(1);
    }
    if (threadIdx.x == 0) __hip_atomic_store((gu64_t *)(g + t), (unsigned long long)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0 && blockIdx.x == (hops - 1) % B) out[0] = wall_clock64() - t0;
}

int main() {
  const int hops = 20000;
  long long *out; unsigned long long *g;
  hipMalloc(&out, 8); hipMalloc(&g, 8 * (size_t)hops);
  long long h;
  for (int W : {2, 4, 16}) {
    for (int v = 0; v < 4; ++v) {
      hipMemset(g, 0xff, 8 * (size_t)hops);
      if (v == 0) hipLaunchKernelGGL(k_lds_chain<0>, dim3(1), dim3(64 * W), 0, 0, hops, out, g);
      if (v == 1) hipLaunchKernelGGL(k_lds_chain<1>, dim3(1), dim3(64 * W), 0, 0, hops, out, g);
      if (v == 2) hipLaunchKernelGGL(k_lds_chain<2>, dim3(1), dim3(64 * W), 0, 0, hops, out, g);
      if (v == 3) hipLaunchKernelGGL(k_lds_chain<3>, dim3(1), dim3(64 * W), 0, 0, hops, out, g);
      hipDeviceSynchronize();
      hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
      printf("LDS chain, %2d waves, %s%s: %.3f us per hop\n", W, (v & 1) ? "sleeping pollers" : "tight pollers", (v & 2) ? " + global sc1 store" : "", h * 0.01 / hops);
    }
  }
  for (int B : {2, 8, 9, 32, 128}) {
    hipMemset(g, 0xff, 8 * (size_t)hops);
    hipLaunchKernelGGL(k_glb_chain, dim3(B), dim3(64), 0, 0, hops, out, g);
    hipDeviceSynchronize();
    hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    printf("global chain, %3d workgroups: %.3f us per hop\n", B, h * 0.01 / hops);
  }
  return 0;
}
