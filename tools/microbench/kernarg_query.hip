// How should a 6 KiB query reach a one-launch lookup?  (a) hipMemcpyAsync from pinned memory in front of the kernel (what libtavb does),
// (b) inside the kernel arguments (6144 bytes by value: every wave reads its copy out of the kernarg segment), (c) read by the kernel straight
// from pinned host memory.  204 workgroups x 16 waves, every wave reads the whole query once (like tavb::scan_fixed_kernel) and reduces it.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/kernarg_query.hip -o tools/microbench/kernarg_query && tools/microbench/kernarg_query
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

struct Query {
  float v[1536];
};

__device__ __forceinline__ float wave_read(const float* q) {
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int i = lane * 4; i < 1536; i += 256) {
    const float4 x = *reinterpret_cast<const float4*>(q + i);
    s += x.x + x.y + x.z + x.w;
  }
  return s;
}

__global__ void __launch_bounds__(1024) from_buffer(const float* q, float* out) {
  const float s = wave_read(q);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = s;
}

__global__ void __launch_bounds__(1024) from_kernarg(const Query q, float* out) {
  const float s = wave_read(q.v);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = s;
}

template <class F>
double median_us(F f, int n = 2000) {
  for (int i = 0; i < 100; ++i) f();
  std::vector<double> t(n);
  for (int i = 0; i < n; ++i) {
    auto a = std::chrono::steady_clock::now();
    f();
    t[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
  }
  std::sort(t.begin(), t.end());
  return t[n / 2];
}

int main() {
  hipStream_t st;
  hipStreamCreate(&st);
  float *d_q, *d_out, *h_q, *h_nc;
  hipMalloc(&d_q, sizeof(Query));
  hipMalloc(&d_out, 204 * 16 * 4);
  hipHostMalloc(&h_q, sizeof(Query), hipHostMallocDefault);
  hipHostMalloc(&h_nc, sizeof(Query), hipHostMallocNonCoherent);
  Query q;
  for (int i = 0; i < 1536; ++i) q.v[i] = h_q[i] = h_nc[i] = 1.0f / (1 + i);
  const dim3 grid(204), block(1024);
  printf("sync only (idle stream):                         %6.1f us\n", median_us([&] { hipStreamSynchronize(st); }));
  printf("kernel (device buffer) + sync:                    %6.1f us\n", median_us([&] { hipLaunchKernelGGL(from_buffer, grid, block, 0, st, d_q, d_out); hipStreamSynchronize(st); }));
  printf("(a) memcpyAsync 6 KiB + kernel + sync:            %6.1f us\n",
         median_us([&] { hipMemcpyAsync(d_q, h_q, sizeof(Query), hipMemcpyHostToDevice, st); hipLaunchKernelGGL(from_buffer, grid, block, 0, st, d_q, d_out); hipStreamSynchronize(st); }));
  printf("(b) kernel with the query in its arguments + sync: %6.1f us\n",
         median_us([&] { q.v[0] += 1.0f; hipLaunchKernelGGL(from_kernarg, grid, block, 0, st, q, d_out); hipStreamSynchronize(st); }));
  printf("(c) kernel reading pinned host memory + sync:      %6.1f us\n", median_us([&] { h_q[0] += 1.0f; hipLaunchKernelGGL(from_buffer, grid, block, 0, st, h_q, d_out); hipStreamSynchronize(st); }));
  printf("(c') ... non-coherent pinned host memory:          %6.1f us\n", median_us([&] { h_nc[0] += 1.0f; hipLaunchKernelGGL(from_buffer, grid, block, 0, st, h_nc, d_out); hipStreamSynchronize(st); }));
  // does (b) deliver fresh data every launch?
  float ref = 0.f, got = 0.f;
  for (int i = 0; i < 1536; ++i) ref += q.v[i];
  hipLaunchKernelGGL(from_kernarg, grid, block, 0, st, q, d_out);
  hipMemcpy(&got, d_out + 203 * 16 + 15, 4, hipMemcpyDeviceToHost);
  printf("kernarg data check: got %.6f expected %.6f\n", got, ref);
  float gotc = 0.f, refc = 0.f;
  for (int i = 0; i < 1536; ++i) refc += h_nc[i];
  hipLaunchKernelGGL(from_buffer, grid, block, 0, st, h_nc, d_out);
  hipMemcpy(&gotc, d_out + 100 * 16, 4, hipMemcpyDeviceToHost);
  printf("non-coherent host data check: got %.6f expected %.6f\n", gotc, refc);
  return 0;
}
