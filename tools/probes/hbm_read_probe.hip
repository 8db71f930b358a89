// Probe (developer tool): what does a kernel that does nothing but read HBM reach on this MI355X?  The decode path's roofline is priced against
// the 8 TB/s peak of the HBM3E stacks; this is the ceiling a perfect streaming read gets in practice — the context for `roofline.frac` of the
// short-query kernels (profiles/r04_hbm_read_probe.txt).  Every lane reads 16 bytes per load (global_load_dwordx4), UNROLL independent loads in
// flight per lane, each workgroup walks a contiguous span (like a KV split walks its keys); the buffer (2 GiB) is 8 x the 256 MB MALL.
//
//   hipcc --offload-arch=gfx950 -O2 tools/probes/hbm_read_probe.hip -o tools/probes/bin/hbm_read_probe && tools/probes/bin/hbm_read_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define CHECK(x)                                                                             \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) {                                                                  \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                               \
    }                                                                                        \
  } while (0)

// CONTIG: each workgroup owns one contiguous span of the buffer (a KV split); otherwise workgroups interleave 4 KiB x UNROLL chunks (a grid-stride copy)
template <int UNROLL, bool CONTIG>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ src, size_t n_vec, unsigned* sink) {
  const size_t per_wg = n_vec / gridDim.x;
  u32x4 acc = {0, 0, 0, 0};
  if (CONTIG) {
    const u32x4* p = src + (size_t)blockIdx.x * per_wg + threadIdx.x;
    for (size_t i = 0; i + 256 * UNROLL <= per_wg; i += 256 * UNROLL) {
      u32x4 t[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) t[u] = __builtin_nontemporal_load(p + i + 256 * u);
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc ^= t[u];
    }
  } else {
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    for (size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i + 256 * (UNROLL - 1) < n_vec; i += stride) {
      u32x4 t[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) t[u] = __builtin_nontemporal_load(src + i + 256 * u);
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc ^= t[u];
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
}

// The same walk by LDS-DMA (buffer_load_dwordx4 ... lds: what the attention kernels stream K / V with): every wave keeps DEPTH 1 KiB pieces in flight
// into a ring of LDS slots (s_waitcnt vmcnt(DEPTH - 1) before a slot is reused); the workgroup's span is contiguous, wave w takes every fourth KiB.
// HINT: 0 = no cache hint, 1 = nt, 2 = sc1, 3 = sc0 sc1 nt.
#define LDSAS __attribute__((address_space(3)))
template <int DEPTH, int HINT>
__global__ __launch_bounds__(256) void dma_kernel(const char* __restrict__ src, size_t bytes, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t per_wg = bytes / gridDim.x;
  const char* base = src + (size_t)blockIdx.x * per_wg;
  const uint64_t ba = (uint64_t)base;
  const u32x4 rsrc = {(uint32_t)ba, (uint32_t)(ba >> 32) & 0xffffu, (uint32_t)per_wg, 0x00020000u};
  const uint32_t lds0 = (uint32_t)(uintptr_t)(LDSAS char*)smem + (uint32_t)wave * DEPTH * 1024u;
  const uint32_t voff = (uint32_t)lane * 16u;
  const uint32_t n_pieces = (uint32_t)(per_wg / 4096);  // per wave
  uint32_t soff = (uint32_t)wave * 1024u;
  for (uint32_t i = 0; i < n_pieces; ++i) {
    const uint32_t dst = lds0 + (i % DEPTH) * 1024u;
    if (HINT == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    if (HINT == 1) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" : : "s"(dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    if (HINT == 2) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc1 lds" : : "s"(dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    if (HINT == 3) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc0 sc1 nt lds" : : "s"(dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    soff += 4096u;
    if (i + 1 >= DEPTH) {
      if (DEPTH == 4) __builtin_amdgcn_s_waitcnt(0x0F70 | 3);
      if (DEPTH == 8) __builtin_amdgcn_s_waitcnt(0x0F70 | 7);
      if (DEPTH == 16) __builtin_amdgcn_s_waitcnt(0x0F70 | 15);
      if (DEPTH == 32) __builtin_amdgcn_s_waitcnt(0x0F70 | (31 & 15) | ((31 >> 4) << 14));
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  if (((LDSAS unsigned*)smem)[threadIdx.x] == 0x12345u) sink[0] = 1;
}

template <int DEPTH, int HINT>
static void run_dma(const char* name, const char* src, size_t bytes, int wgs, unsigned* sink) {
  auto k = dma_kernel<DEPTH, HINT>;
  const int lds = 4 * DEPTH * 1024;
  CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(k, dim3(wgs), dim3(256), lds, 0, src, bytes, sink);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int reps = 20;
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(wgs), dim3(256), lds, 0, src, bytes, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("HBMDMA  %-44s %5d workgroups x 4 waves, %2d KiB pieces in flight per wave (%3d KiB LDS) | %7.3f ms | %6.2f TB/s (%.3f of 8)\n", name, wgs, DEPTH, lds / 1024, ms,
         bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 8e12);
  fflush(stdout);
}

template <int UNROLL, bool CONTIG>
static void run(const char* name, const u32x4* src, size_t bytes, int wgs, unsigned* sink) {
  const size_t n_vec = bytes / 16;
  hipLaunchKernelGGL((read_kernel<UNROLL, CONTIG>), dim3(wgs), dim3(256), 0, 0, src, n_vec, sink);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int reps = 20;
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((read_kernel<UNROLL, CONTIG>), dim3(wgs), dim3(256), 0, 0, src, n_vec, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("HBMREAD %-44s %5d workgroups x 256 lanes, %2d loads in flight per lane | %7.3f ms | %6.2f TB/s (%.3f of 8)\n", name, wgs, UNROLL, ms,
         bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 8e12);
  fflush(stdout);
}

int main() {
  const size_t bytes = 2ull << 30;
  u32x4* src;
  unsigned* sink;
  CHECK(hipMalloc(&src, bytes));
  CHECK(hipMemset(src, 0x5a, bytes));
  CHECK(hipMalloc(&sink, 64));
  for (int pass = 0; pass < 2; ++pass) {
    run<4, true>("contiguous span per workgroup", src, bytes, 256, sink);
    run<4, true>("contiguous span per workgroup", src, bytes, 512, sink);
    run<8, true>("contiguous span per workgroup", src, bytes, 512, sink);
    run<8, true>("contiguous span per workgroup", src, bytes, 1024, sink);
    run<8, true>("contiguous span per workgroup", src, bytes, 2048, sink);
    run<16, true>("contiguous span per workgroup", src, bytes, 2048, sink);
    run<8, true>("contiguous span per workgroup", src, bytes, 8192, sink);
    run<8, false>("interleaved chunks (grid-stride)", src, bytes, 2048, sink);
    run<8, false>("interleaved chunks (grid-stride)", src, bytes, 8192, sink);
    run<16, false>("interleaved chunks (grid-stride)", src, bytes, 4096, sink);
    run<8, true>("contiguous span per workgroup", src, bytes, 256, sink);
    run<16, true>("contiguous span per workgroup", src, bytes, 256, sink);
    const char* cs = (const char*)src;
    run_dma<4, 0>("LDS-DMA, no hint", cs, bytes, 256, sink);
    run_dma<8, 0>("LDS-DMA, no hint", cs, bytes, 256, sink);
    run_dma<16, 0>("LDS-DMA, no hint", cs, bytes, 256, sink);
    run_dma<32, 0>("LDS-DMA, no hint", cs, bytes, 256, sink);
    run_dma<8, 0>("LDS-DMA, no hint", cs, bytes, 512, sink);
    run_dma<16, 0>("LDS-DMA, no hint", cs, bytes, 512, sink);
    run_dma<8, 0>("LDS-DMA, no hint", cs, bytes, 1024, sink);
    run_dma<16, 1>("LDS-DMA, nt", cs, bytes, 256, sink);
    run_dma<16, 1>("LDS-DMA, nt", cs, bytes, 512, sink);
    run_dma<16, 2>("LDS-DMA, sc1", cs, bytes, 256, sink);
    run_dma<16, 3>("LDS-DMA, sc0 sc1 nt", cs, bytes, 256, sink);
    run_dma<16, 3>("LDS-DMA, sc0 sc1 nt", cs, bytes, 512, sink);
  }
  return 0;
}
