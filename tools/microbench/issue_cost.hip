// Micro-benchmark: what does it cost a wave that issues MFMAs back to back (one wave per SIMD, like MFMA variant 5) to
// slot operand loads between them?  Per iteration: 24 independent 32x32x16 f16 MFMAs (768 matrix-pipe cycles) and 5 loads
// of 1 KiB per wave, issued one per ~5 MFMAs.
//   mode 0  MFMAs only
//   mode 1  + 5 x buffer_load_dwordx4 ... lds  (LDS-DMA)
//   mode 4  + 10 x ds_read_b128 (fragment reads) but no loads
//   mode 5  + 10 x ds_read_b128 and 5 x LDS-DMA (what the MFMA kernel's K loop does)
// All 256 CUs run; the loads stream through a window that is L2 resident (768 KiB) or not (8 GiB).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -o issue_cost issue_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

template <int MODE>
__global__ __launch_bounds__(256) void k(const char* __restrict__ src, size_t window, int iters, float* sink, int rows_pattern) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x16 acc[24];
#pragma unroll
  for (int i = 0; i < 24; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  i32x4 a = {lane, 1, 2, 3}, b = {4, 5, 6, lane};
  // rows_pattern: a piece = 16 rows x 64 bytes of a 3072-byte-pitch matrix (what the MFMA kernels stage), 48 K steps along
  // the rows before moving to the next 80-row block; otherwise a piece = 1 KiB contiguous.
  // rows_pattern 2: a piece = 8 rows x 128 bytes (whole cache lines) of the same matrix, 24 K steps along the rows.
  const int voff = rows_pattern == 2 ? (lane >> 3) * 3072 + (lane & 7) * 16 : rows_pattern ? (lane >> 2) * 3072 + (lane & 3) * 16 : lane * 16;
  const int piece = rows_pattern == 2 ? 8 * 3072 : rows_pattern ? 16 * 3072 : 1024;
  const size_t blk = rows_pattern == 2 ? (size_t)40 * 3072 : rows_pattern ? (size_t)80 * 3072 : 5120;
  const int kstep = rows_pattern == 2 ? 128 : 64, ksteps = rows_pattern == 2 ? 24 : 48;
  const size_t nblk = window / blk;
  size_t rb = ((size_t)blockIdx.x * 4 + wave) % nblk;
  int kt = 0;
  i32x4 fr[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) fr[j] = i32x4{0, 0, 0, 0};
  unsigned char* my_lds = smem + wave * 5120;
  for (int it = 0; it < iters; ++it) {
    const char* g = src + rb * blk + (rows_pattern ? kt * kstep : 0);
    const unsigned lo = __builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)g);
    const unsigned hi = __builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)g >> 32));
    const char* ug = (const char*)(((unsigned long long)hi << 32) | lo);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ug, 0, (int)blk, 0x00020000);
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      if (i < 15)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
      else
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      if (i % 5 == 3) {
        const int j = i / 5;
        if (MODE == 1 || MODE == 5) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(my_lds + j * 1024), 16, voff, j * piece, 0, 0);
      }
      if ((MODE == 4 || MODE == 5) && i < 10) {
        const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(smem + 20480 + ((i + wave) & 7) * 1024) + lane * 16;
        asm volatile("ds_read_b128 %0, %1" : "=v"(fr[i]) : "v"(la) : "memory");
      }
    }
    if (MODE == 4 || MODE == 5) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 10; ++i) a.x ^= fr[i].x & 1;
    }
    if (MODE == 1 || MODE == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    if (rows_pattern && ++kt < ksteps) continue;
    kt = 0;
    rb += (size_t)gridDim.x * 4;
    while (rb >= nblk) rb -= nblk;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  float s = 0;
#pragma unroll
  for (int i = 0; i < 24; ++i) s += acc[i][0];
#pragma unroll
  for (int j = 0; j < 10; ++j) s += (float)fr[j].x;
  if (s == 12345.678f) sink[0] = s;
}

template <int MODE>
static void run(const char* name, const char* d, size_t window, int iters, float* sink, int cus, int pat) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<cus, 256, 32768>>>(d, window, iters / 8, sink, pat);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<MODE><<<cus, 256, 32768>>>(d, window, iters, sink, pat);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s %s window %8.1f MiB: %8.3f ms  %7.1f ns per iteration (24 MFMAs + 5 KiB per wave)  %6.1f TFLOP/s-eq\n", name, pat == 2 ? "rows8x128B" : pat ? "rows16x64B" : "contiguous", window / 1048576.0, ms,
         ms * 1e6 / iters, 24.0 * 32768 * 4 * cus * iters / (ms * 1e-3) / 1e12);
}

int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const size_t big = (size_t)8 << 30;
  char* d; float* sink;
  (void)hipMalloc(&d, big + (1 << 20));
  (void)hipMemset(d, 1, big + (1 << 20));
  (void)hipMalloc(&sink, 4);
  const int iters = 20000;
  for (size_t window : {(size_t)983040, (size_t)64 << 20, big}) {
    for (int pat = 0; pat < 3; ++pat) {
      if (pat == 0 && window != 983040) continue;
      run<0>("MFMAs only", d, window, iters, sink, cus, pat);
      run<1>("+ 5 LDS-DMA pieces", d, window, iters, sink, cus, pat);
      run<5>("+ 10 ds_read_b128 + 5 LDS-DMA", d, window, iters, sink, cus, pat);
    }
  }
  return 0;
}
