// Microbenchmark: latency of a producer -> consumer hand-off between workgroups of ONE launch on gfx950 (the decoupled look-back
// idiom k_bwd_chain uses) against a kernel boundary inside a captured hipGraph.  A chain of N workgroups: workgroup b waits until
// the counter reaches b, reads the 96 doubles workgroup b-1 wrote, adds 1, writes its own 96 doubles, publishes.
//   mode 0: plain stores + agent release fence | plain loads after an agent acquire fence
//   mode 1: plain stores + agent release fence | agent-scope atomic loads (no acquire fence)
//   mode 2: agent-scope atomic stores + workgroup release (store acknowledge) | agent-scope atomic loads
// and N single-workgroup kernels replayed from a hipGraph doing the same step (plain loads / stores).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ __launch_bounds__(1024) void k_chain(double *buf, unsigned *done, int mode, int stride) {
  const int b = blockIdx.x;
  if (b > 0) {
    if (threadIdx.x == 0) while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)b) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
    if (mode == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  if (threadIdx.x < 96) {
    double v = 0.0;
    if (b > 0) {
      const double *src = buf + (size_t)(b - 1) * stride + threadIdx.x;
      v = mode == 0 ? *src : __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    double *dst = buf + (size_t)b * stride + threadIdx.x;
    if (mode == 2) __hip_atomic_store(dst, v + 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *dst = v + 1.0;
  }
  if (threadIdx.x < 64) {                                 // (wave 0 holds lanes 0..63; lanes 64..95 are wave 1: publish after both)
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (mode == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ __launch_bounds__(1024) void k_step(double *buf, int b, int stride) {
  if (threadIdx.x < 96) {
    const double v = b > 0 ? buf[(size_t)(b - 1) * stride + threadIdx.x] : 0.0;
    buf[(size_t)b * stride + threadIdx.x] = v + 1.0;
  }
}
int main() {
  const int N = 512, stride = 128;
  double *buf; unsigned *done;
  CHK(hipMalloc(&buf, sizeof(double) * N * stride)); CHK(hipMalloc(&done, 4));
  hipStream_t s; CHK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int mode = 0; mode < 3; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      CHK(hipMemsetAsync(done, 0, 4, s)); CHK(hipMemsetAsync(buf, 0, sizeof(double) * N * stride, s));
      CHK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(k_chain, dim3(N), dim3(1024), 0, s, buf, done, mode, stride);
      CHK(hipEventRecord(e1, s)); CHK(hipStreamSynchronize(s));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
      double last[96]; CHK(hipMemcpy(last, buf + (size_t)(N - 1) * stride, sizeof(last), hipMemcpyDeviceToHost));
      if (rep == 2) std::printf("chain of %d workgroups, mode %d: %.1f us total, %.2f us per hand-off (check %.0f == %d)\n", N, mode, 1e3 * ms, 1e3 * ms / N, last[5], N);
    }
  hipGraph_t g; hipGraphExec_t ge;
  CHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int b = 0; b < N; ++b) hipLaunchKernelGGL(k_step, dim3(1), dim3(1024), 0, s, buf, b, stride);
  CHK(hipStreamEndCapture(s, &g)); CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int rep = 0; rep < 3; ++rep) {
    CHK(hipEventRecord(e0, s)); CHK(hipGraphLaunch(ge, s)); CHK(hipEventRecord(e1, s)); CHK(hipStreamSynchronize(s));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 2) std::printf("%d single-workgroup kernels from a hipGraph: %.1f us total, %.2f us per kernel\n", N, 1e3 * ms, 1e3 * ms / N);
  }
  return 0;
}
