// What does hipStreamSynchronize cost behind a one-launch lookup, against a completion flag the LAST workgroup writes into pinned host memory and the
// host spins on?  204 workgroups x 16 waves, every wave reads a 6 KiB query and writes one float into pinned host memory (like the per-workgroup
// lists of libtavb's small-corpus path).
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/flag_completion.hip -o tools/microbench/flag_completion && tools/microbench/flag_completion
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
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

__global__ void __launch_bounds__(1024) plain(const float* q, float* out) {
  const float s = wave_read(q);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = s;
}

__device__ __forceinline__ void finish(unsigned* counter, unsigned* flag, unsigned seq) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    const unsigned t = atomicAdd(counter, 1u);
    if (t == gridDim.x - 1) {
      *counter = 0;
      __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void __launch_bounds__(1024) flagged(const float* q, float* out, unsigned* counter, unsigned* flag, unsigned seq) {
  const float s = wave_read(q);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = s + (float)seq;
  finish(counter, flag, seq);
}

__global__ void __launch_bounds__(1024) flagged_kernarg(const Query q, float* out, unsigned* counter, unsigned* flag, unsigned seq) {
  const float s = wave_read(q.v);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = s + (float)seq;
  finish(counter, flag, seq);
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
  float *d_q, *d_out, *h_q, *h_out;
  unsigned *d_counter, *h_flag;
  hipMalloc(&d_q, sizeof(Query));
  hipMalloc(&d_out, 204 * 16 * 4);
  hipMalloc(&d_counter, 4);
  hipMemset(d_counter, 0, 4);
  hipHostMalloc(&h_q, sizeof(Query), hipHostMallocDefault);
  hipHostMalloc(&h_out, 204 * 16 * 4, hipHostMallocDefault);
  hipHostMalloc(&h_flag, 64, hipHostMallocDefault);
  *h_flag = 0;
  Query q;
  for (int i = 0; i < 1536; ++i) q.v[i] = h_q[i] = 1.0f / (1 + i);
  hipMemcpy(d_q, h_q, sizeof(Query), hipMemcpyHostToDevice);
  const dim3 grid(204), block(1024);
  unsigned seq = 0;
  long bad = 0;
  auto spin = [&](unsigned want) {
    volatile unsigned* f = h_flag;
    long spins = 0;
    while (*f != want) {
      __builtin_ia32_pause();
      if (++spins > 2000000000L) {
        printf("flag never arrived (want %u, have %u)\n", want, *f);
        exit(3);
      }
    }
  };
  printf("kernel -> device buffer, sync:                                 %6.1f us\n", median_us([&] { hipLaunchKernelGGL(plain, grid, block, 0, st, d_q, d_out); hipStreamSynchronize(st); }));
  printf("kernel -> pinned host lists, sync:                             %6.1f us\n", median_us([&] { hipLaunchKernelGGL(plain, grid, block, 0, st, d_q, h_out); hipStreamSynchronize(st); }));
  printf("copy 6 KiB + kernel -> pinned lists, sync  (libtavb today):    %6.1f us\n",
         median_us([&] { hipMemcpyAsync(d_q, h_q, sizeof(Query), hipMemcpyHostToDevice, st); hipLaunchKernelGGL(plain, grid, block, 0, st, d_q, h_out); hipStreamSynchronize(st); }));
  printf("kernel -> pinned lists + flag, sync (flag unused):             %6.1f us\n",
         median_us([&] { ++seq; hipLaunchKernelGGL(flagged, grid, block, 0, st, d_q, h_out, d_counter, h_flag, seq); hipStreamSynchronize(st); }));
  printf("kernel -> pinned lists + flag, host spins on the flag:         %6.1f us\n", median_us([&] {
           ++seq;
           hipLaunchKernelGGL(flagged, grid, block, 0, st, d_q, h_out, d_counter, h_flag, seq);
           spin(seq);
           if (h_out[203 * 16 + 15] != h_out[0]) ++bad;   // every list must already be there when the flag is
         }));
  printf("copy 6 KiB + kernel + flag, host spins:                        %6.1f us\n", median_us([&] {
           ++seq;
           hipMemcpyAsync(d_q, h_q, sizeof(Query), hipMemcpyHostToDevice, st);
           hipLaunchKernelGGL(flagged, grid, block, 0, st, d_q, h_out, d_counter, h_flag, seq);
           spin(seq);
         }));
  printf("query in the kernel arguments + flag, host spins:              %6.1f us\n", median_us([&] {
           ++seq;
           q.v[0] += 1.0f;
           hipLaunchKernelGGL(flagged_kernarg, grid, block, 0, st, q, h_out, d_counter, h_flag, seq);
           spin(seq);
         }));
  // all lists visible with the flag?  check every slot of 20000 launches against the sequence number folded into them
  float base = 0.f;
  for (int i = 0; i < 1536; ++i) base += h_q[i];
  for (int it = 0; it < 20000; ++it) {
    ++seq;
    hipLaunchKernelGGL(flagged, grid, block, 0, st, d_q, h_out, d_counter, h_flag, seq);
    spin(seq);
    const float want = h_out[0];
    for (int i = 0; i < 204 * 16; ++i)
      if (h_out[i] != want) { ++bad; break; }
  }
  hipStreamSynchronize(st);
  printf("launches whose lists were NOT all visible when the flag was: %ld\n", bad);
  return 0;
}
