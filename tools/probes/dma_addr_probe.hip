// Probe (developer tool): does an LDS-DMA piece cost its wave less issue time when the per-lane source address comes from the
// buffer descriptor (ADD_TID_ENABLE: address = base + soffset + 16 * lane, no offset VGPR) instead of an `offen` VGPR?
//
// The attention kernels pay 45 - 60 cycles of an in-order wave per 1 KiB piece (`buffer_load_dwordx4 v_off, rsrc, s_off offen lds`)
// issued among MFMAs.  If the VGPR operand were part of that price, the kernels' source-side swizzle (which is what needs per-lane
// offsets) could move to the read side.  Measured here: (1) both forms copy the same bytes, (2) cycles per loop iteration of
// [NM dependent-free MFMAs + 1 piece] for NM = 0, 2, 4, 8 with 4 waves per CU on all 256 CUs.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/dma_addr_probe.hip -o tools/probes/bin/dma_addr_probe && tools/probes/bin/dma_addr_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define LDSAS __attribute__((address_space(3)))

#define CHECK(x)                                                                             \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) {                                                                  \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                               \
    }                                                                                        \
  } while (0)

struct Args {
  const char* src;
  uint32_t region;  // bytes per XCD slice
  int iters;
  uint32_t* out;    // check mode: LDS image copied back
  float* sink;
  unsigned long long* ticks;
};

// MODE 0: offen VGPR offset; MODE 1: descriptor adds 16 * lane (stride 16, ADD_TID_ENABLE), no VGPR
template <int MODE>
__device__ __forceinline__ void piece(u32x4 rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  if constexpr (MODE == 0)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 off, %1, %2 lds" : : "s"(lds_addr), "s"(rsrc), "s"(soff) : "memory");
}

template <int MODE>
__device__ __forceinline__ u32x4 make_desc(const char* base, uint32_t bytes) {
  const uint64_t ba = (uint64_t)base;
  if constexpr (MODE == 0) return u32x4{(uint32_t)ba, (uint32_t)(ba >> 32) & 0xffffu, bytes, 0x00020000u};
  // stride 16 bytes in word1[29:16]; num_records counts strides when stride != 0; ADD_TID_ENABLE = word3 bit 23
  // (with ADD_TID_ENABLE the DATA_FORMAT field of word 3 holds stride[17:14]: it must be 0 here)
  return u32x4{(uint32_t)ba, ((uint32_t)(ba >> 32) & 0xffffu) | (16u << 16), bytes / 16u, (1u << 23)};
}

template <int MODE>
__global__ __launch_bounds__(256) void check_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32x4 rsrc = make_desc<MODE>(a.src, a.region);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(LDSAS char*)smem;
  for (int i = threadIdx.x; i < 4096; i += 256) ((LDSAS uint32_t*)smem)[i] = 0xdeadbeefu;
  __syncthreads();
  // 4 pieces per wave: piece p of wave w <- source bytes [(4 w + p) KiB + 512, ...) (a non-zero scalar offset on purpose)
  for (int p = 0; p < 4; ++p) piece<MODE>(rsrc, lds0 + (uint32_t)((wave * 4 + p) * 1024), (uint32_t)lane * 16u, (uint32_t)((wave * 4 + p) * 1024 + 512));
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += 256) a.out[i] = ((LDSAS uint32_t*)smem)[i];
}

template <int MODE, int NM>
__global__ __launch_bounds__(256) void rate_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7;
  const u32x4 rsrc = make_desc<MODE>(a.src + (size_t)xcd * a.region, a.region);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(LDSAS char*)smem;
  const uint32_t voff = (uint32_t)lane * 16u;
  f32x4 acc[NM > 0 ? NM : 1];
#pragma unroll
  for (int i = 0; i < (NM > 0 ? NM : 1); ++i) acc[i] = (f32x4)(0.f);
  const bf16x8 x = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u + lane, 0x3f813f7fu, 0x3f7e3f82u, 0x3f803f80u});
  uint32_t soff = (uint32_t)wave * 16384u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < a.iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, acc[i], 0, 0, 0);
      piece<MODE>(rsrc, lds0 + (uint32_t)(((wave * 16 + u) * 1024) & (128 * 1024 - 1)), voff, soff + (uint32_t)u * 1024u);
    }
    __builtin_amdgcn_sched_barrier(0);
    soff += 65536u;
    if (soff + 65536u > a.region) soff = (uint32_t)wave * 16384u;
    __builtin_amdgcn_s_waitcnt(0x0F70 | (16 & 15) | ((16 >> 4) << 14));  // at most the youngest 16 pieces in flight
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < (NM > 0 ? NM : 1); ++i) s += acc[i][0];
  if (s == 12345.678f) a.sink[0] = s;
  if (lane == 0) a.ticks[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE, int NM>
static void run_rate(Args a, int iters) {
  a.iters = iters;
  auto k = rate_kernel<MODE, NM>;
  const int lds = 128 * 1024;
  CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(k, dim3(256), dim3(256), lds, 0, a);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(256), lds, 0, a);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 3;
  std::vector<unsigned long long> h(256 * 4);
  CHECK(hipMemcpy(h.data(), a.ticks, h.size() * 8, hipMemcpyDeviceToHost));
  double avg = 0;
  for (auto t : h) avg += (double)t;
  avg /= h.size();
  const double pieces = 16.0 * iters;
  printf("RATE %-26s NM %d | %7.3f ms | %6.1f shader cycles per [%d MFMA + 1 piece] per wave | LDS-DMA %5.2f TB/s chip | MFMA %6.1f TFLOP/s\n",
         MODE == 0 ? "offen VGPR offset" : "descriptor adds 16*lane", NM, ms, avg / pieces, NM, pieces * 1024.0 * 1024.0 / (ms * 1e-3) / 1e12,
         pieces * NM * 16384.0 * 1024.0 / (ms * 1e-3) / 1e12);
  fflush(stdout);
}

int main() {
  const uint32_t region = 32u << 20;
  char* src;
  CHECK(hipMalloc(&src, (size_t)region * 8));
  std::vector<uint32_t> h((size_t)region * 8 / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u) ^ 0x5bd1e995u;
  CHECK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  Args a{};
  a.src = src;
  a.region = region;
  CHECK(hipMalloc(&a.out, 16384));
  CHECK(hipMalloc(&a.sink, 64));
  CHECK(hipMalloc(&a.ticks, 256 * 4 * 8));
  // (1) both forms copy the same bytes
  for (int mode = 0; mode < 2; ++mode) {
    const int lds = 16384;
    if (mode == 0) hipLaunchKernelGGL(check_kernel<0>, dim3(1), dim3(256), lds, 0, a);
    else hipLaunchKernelGGL(check_kernel<1>, dim3(1), dim3(256), lds, 0, a);
    CHECK(hipDeviceSynchronize());
    std::vector<uint32_t> o(4096);
    CHECK(hipMemcpy(o.data(), a.out, 16384, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int p = 0; p < 16; ++p)
      for (int w = 0; w < 256; ++w)
        if (o[p * 256 + w] != h[(p * 1024 + 512) / 4 + w]) ++bad;
    printf("CHECK %-26s: %d of 4096 dwords differ from the source image\n", mode == 0 ? "offen VGPR offset" : "descriptor adds 16*lane", bad);
    if (bad != 0) {
      printf("the descriptor form does not copy the same bytes: no rate runs\n");
      return 0;
    }
  }
  // (2) issue cost
  const int iters = 2048;
  for (int rep = 0; rep < 2; ++rep) {
    run_rate<0, 0>(a, iters);
    run_rate<1, 0>(a, iters);
    run_rate<0, 2>(a, iters);
    run_rate<1, 2>(a, iters);
    run_rate<0, 4>(a, iters);
    run_rate<1, 4>(a, iters);
    run_rate<0, 8>(a, iters);
    run_rate<1, 8>(a, iters);
  }
  return 0;
}
