// Micro-benchmark: per-CU throughput of the three ways a 1 KiB wave-piece can reach a CU from L2 on gfx950:
//   mode 0  global_load_lds_dwordx4 (LDS-DMA)            -- what the MFMA kernel's stagers use
//   mode 1  global_load_dwordx4 into VGPRs (consumed by an xor)
//   mode 2  global_load_dwordx4 into VGPRs, then ds_write_b128 (register staging)
//   mode 3  half the pieces by LDS-DMA, half by VGPR + ds_write (are the two paths independent?)
// One 512-thread workgroup per CU, every wave streams `iters` rounds of 4 pieces from a window of
// `window` bytes that all workgroups share (L2-resident, far larger than the 32 KiB L1), like the query tile.
// Build: hipcc --offload-arch=gfx950 -O3 -o load_paths load_paths.hip ; run: ./load_paths
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32;
typedef __attribute__((ext_vector_type(4))) u32 u32x4;

template <int MODE>
__global__ __launch_bounds__(512) void load_kernel(const char* __restrict__ src, size_t window, int iters, u32* sink, int rows_pattern) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char* my_lds = lds + wave * 4096;  // 4 pieces of 1 KiB per wave
  u32x4 acc = {0, 0, 0, 0};
  const u32 lds_addr = (u32)(size_t)(__attribute__((address_space(3))) char*)my_lds;
  size_t off = ((size_t)blockIdx.x * 8 + wave) * 4096 % window;
  // rows_pattern: a piece = 16 rows x 64 B of a 3072-byte-pitch matrix (what the MFMA kernel stages), walking 48 column
  // chunks before moving to the next 64-row block; otherwise a piece = 1 KiB contiguous.
  const size_t step = ((size_t)gridDim.x * 8 * 4096) % window;
  const int nrb = (int)(window / (64 * 3072));
  int rb = (blockIdx.x * 8 + wave) % (nrb > 0 ? nrb : 1), chunk = 0;
  // rows_pattern 2: a piece = 8 rows x 128 B (whole cache lines), 24 column chunks per 32-row block (x2 blocks = the same 64 rows).
  const size_t lane_off = rows_pattern == 2 ? (size_t)(lane >> 3) * 3072 + (lane & 7) * 16 : rows_pattern ? (size_t)(lane >> 2) * 3072 + (lane & 3) * 16 : (size_t)lane * 16;
  const size_t piece = rows_pattern == 2 ? (size_t)8 * 3072 : rows_pattern ? (size_t)16 * 3072 : 1024;
  const int kbytes = rows_pattern == 2 ? 128 : 64, kchunks = rows_pattern == 2 ? 24 : 48;
  auto base = [&]() -> const char* {
    return rows_pattern ? src + (size_t)rb * 64 * 3072 + (size_t)chunk * kbytes + lane_off : src + off + lane_off;
  };
  auto advance = [&]() {
    if (rows_pattern) {
      if (++chunk == kchunks) {
        chunk = 0;
        rb = (rb + gridDim.x * 8) % nrb;
      }
    } else {
      off += step;
      if (off >= window) off -= window;
    }
  };
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
      const char* g = base();
#pragma unroll
      for (int p = 0; p < 4; ++p)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + p * piece),
                                         (__attribute__((address_space(3))) void*)(my_lds + p * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // two rounds stay in flight
      advance();
    }
  } else {
    // register paths: two rounds (8 loads) in flight per wave, like the DMA mode
    constexpr int NV = (MODE == 3) ? 2 : 4;
    u32x4 va[NV], vb[NV];
    auto issue = [&](u32x4* v) {
      const char* g = base();
      if (MODE == 3) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + p * piece),
                                           (__attribute__((address_space(3))) void*)(my_lds + p * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int p = 0; p < NV; ++p) v[p] = *(const u32x4*)(g + (4 - NV + p) * piece);
      advance();
    };
    auto consume = [&](u32x4* v) {
      if (MODE == 1) {
#pragma unroll
        for (int p = 0; p < NV; ++p) acc ^= v[p];
      } else {
#pragma unroll
        for (int p = 0; p < NV; ++p)
          asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr + (4 - NV + p) * 1024 + lane * 16), "v"(v[p]) : "memory");
      }
    };
    issue(va);
    for (int it = 0; it < iters; it += 2) {
      issue(vb);
      consume(va);
      issue(va);
      consume(vb);
    }
    consume(va);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (MODE != 1) acc = *(u32x4*)(my_lds + lane * 16);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int MODE>
static void run(const char* name, const char* d, size_t window, int iters, u32* sink, int cus, int rows_pattern) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  load_kernel<MODE><<<cus, 512, 32768>>>(d, window, iters / 8, sink, rows_pattern);
  hipDeviceSynchronize();
  hipEventRecord(a);
  load_kernel<MODE><<<cus, 512, 32768>>>(d, window, iters, sink, rows_pattern);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  double bytes_per_cu = (double)iters * 8 * 4096;
  printf("%-28s %s window %7.1f MiB: %8.3f ms  %7.1f GB/s per CU  %6.2f TB/s chip  (%.1f ns per 1 KiB piece per CU)\n", name, rows_pattern == 2 ? "rows8x128B" : rows_pattern ? "rows16x64B" : "contiguous",
         window / 1048576.0, ms, bytes_per_cu / ms * 1e-6, bytes_per_cu * cus / ms * 1e-9, ms * 1e6 / (iters * 8 * 4.0));
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  int cus = prop.multiProcessorCount;
  size_t maxwin = (size_t)8 << 30;
  char* d; u32* sink;
  hipMalloc(&d, maxwin + (1 << 20));
  hipMemset(d, 1, maxwin + (1 << 20));
  hipMalloc(&sink, 4);
  // 768 KiB = one query tile (L2 hits); 64 MiB = MALL-resident; 8 GiB = HBM stream
  for (size_t window : {(size_t)786432, (size_t)64 << 20, maxwin}) {
    size_t w = window - window % ((size_t)64 * 3072);
    // the window must be a multiple of 4 KiB; offsets wrap by subtraction
    int iters = 20000;
    for (int pat = 0; pat < 3; ++pat) {
      run<0>("lds-dma", d, w, iters, sink, cus, pat);
      run<1>("vgpr", d, w, iters, sink, cus, pat);
      run<2>("vgpr + ds_write", d, w, iters, sink, cus, pat);
      run<3>("half dma, half vgpr+ds_write", d, w, iters, sink, cus, pat);
    }
  }
  return 0;
}
