// Probe (developer tool): what would HALVING the LDS fragment reads of the D = 512 tile buy, and what would the two ways of getting there cost?
//
// The shipped D = 512 tile: four one-wave-per-SIMD waves, each owning 32 query rows x all of D.  Per 64-key tile a wave issues 256 MFMAs
// (16x16x32 bf16), and every K / V^T fragment it reads from LDS (ds_read_b128-sized, 1 KiB per wave) feeds TWO of them (the wave's two 16-row
// halves): 128 fragment reads per wave and tile, all four waves reading the whole K and V tile.  A wave owning 64 rows x D/2 would feed FOUR
// MFMAs from each fragment and read only half of each tile: 64 reads per wave and tile.  It has to get its scores from somewhere, though:
//   (a) split the score GEMM over D like the D > 512 tiles do: each wave of a pair computes the partial S^T of the pair's 64 rows over its
//       half of D, the partials (64 rows x 64 keys fp32 = 16 KiB per wave) cross through LDS behind one more barrier, and BOTH waves of the
//       pair run the softmax of all 64 rows (twice the VALU per wave: rows cannot be divided, each wave needs every P entry for its PV half);
//   (b) split the score GEMM over keys: needs the Q fragments of 64 rows x full D in registers = 256 VGPRs per lane.  Does not exist.
// This probe times the instruction skeleton of one tile (MFMA halves with their fragment reads and LDS-DMA pieces, a VALU stand-in for the
// softmax, three workgroup barriers — the same skeleton as stream_probe.hip's probe_soft) for: the shipped shape; (a); and two bounds that no
// kernel can reach — four MFMAs per fragment in both halves with nothing else changed, and in the PV half only.
//
//   hipcc --offload-arch=gfx950 -O2 tools/probes/share_probe.hip -o tools/probes/bin/share_probe && tools/probes/bin/share_probe
//
// Output recorded in profiles/r04_share_probe.txt.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4p;
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
  const char* src;       // stream image, `region` bytes per XCD slice (L2 resident)
  uint32_t region;
  int tiles;
  const uint32_t* bsrc;  // 64 x 4 dwords: B operands (random bf16)
  float* sink;
};

__device__ __forceinline__ void lds_dma(u32x4 rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// NM16 MFMAs per wave and tile in two halves (QK^T, PV); SH0 / SH1 MFMAs per fragment read in the two halves; NDMA 1 KiB pieces per wave and
// tile spread over the fragments; NV VALU instructions (a quarter of them v_exp_f32) between the halves; XCH bytes of fp32 partials written
// to LDS and as many read back, behind one extra barrier, before the VALU block.
template <int NM16, int NDMA, int SH0, int SH1, int NV, int XCH>
__global__ __launch_bounds__(256) void probe_share(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7;
  const char* base = a.src + (size_t)xcd * a.region;
  const uint64_t ba = (uint64_t)base;
  const u32x4 rsrc = {(uint32_t)ba, (uint32_t)(ba >> 32) & 0xffffu, a.region, 0x00020000u};
  const uint32_t lds0 = (uint32_t)(uintptr_t)(LDSAS char*)smem;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((LDSAS uint32_t*)smem)[i] = 0x3f803f80u ^ (uint32_t)(i * 2654435761u >> 12 & 0x00ff00ffu);
  __syncthreads();
  bf16x8 b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = __builtin_bit_cast(bf16x8, *(const u32x4*)(a.bsrc + ((lane + 7 * i) & 63) * 4));
  f32x4p acc[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = (f32x4p)(0.f);
  float vs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) vs[i] = (float)(lane + i) * 1e-3f;
  const uint32_t voff = (uint32_t)lane * 16u;
  constexpr uint32_t per_tile = (uint32_t)(4 * NDMA) * 1024u;
  uint32_t soff = (uint32_t)wave * (uint32_t)NDMA * 1024u;
  constexpr int PF = 3;
  bf16x8 fr[4];
  auto frag_read = [&](int f) -> bf16x8 {
    const u32x4 raw = *(LDSAS const u32x4*)(smem + ((f * 1024) & 0xffff) + lane * 16);
    return __builtin_bit_cast(bf16x8, raw);
  };
#pragma unroll
  for (int f = 0; f < PF; ++f) fr[f] = frag_read(f);
  auto mfma_half = [&](auto hc) {
    constexpr int h = decltype(hc)::value;
    constexpr int SH = h == 0 ? SH0 : SH1;
    constexpr int NF = NM16 / 2 / SH;          // fragments of this half
    constexpr int ND = NDMA / 2;               // pieces of this half
    constexpr int STEP = NF / ND > 0 ? NF / ND : 1;
    constexpr int PER = ND > NF ? ND / NF : 1;  // pieces per fragment when there are more pieces than fragments
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      __builtin_amdgcn_sched_barrier(0);
      fr[(f + PF) & 3] = frag_read(f + PF + h * 64);
      if (f % STEP == 0 && f / STEP * PER < ND) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          const int piece = h * ND + f / STEP * PER + q;
          lds_dma(rsrc, lds0 + (uint32_t)(((wave * NDMA + piece) * 1024) & (128u * 1024u - 1)), voff, soff + (uint32_t)piece * 1024u);
        }
      }
#pragma unroll
      for (int r = 0; r < SH; ++r) {
        const int i = (f * SH + r + h * 32) & 63;
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[f & 3], b[r & 3], acc[i], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
  };
  auto exchange = [&]() {
    if constexpr (XCH > 0) {
      // partial scores out (one 16-byte store per lane and KiB) ...
      LDSAS char* mine = (LDSAS char*)smem + 128 * 1024 - 4 * XCH + wave * XCH;
      LDSAS const char* theirs = (LDSAS const char*)smem + 128 * 1024 - 4 * XCH + (wave ^ 1) * XCH;
#pragma unroll
      for (int i = 0; i < XCH / 1024; ++i) *(LDSAS f32x4p*)(mine + i * 1024 + lane * 16) = acc[i & 63];
      bar();
      // ... and the partner's in, added to the wave's own
#pragma unroll
      for (int i = 0; i < XCH / 1024; ++i) {
        const f32x4p p = *(LDSAS const f32x4p*)(theirs + i * 1024 + lane * 16);
        vs[i & 7] += p[0] + p[1] + p[2] + p[3];
      }
    }
  };
  auto valu = [&]() {
#pragma unroll
    for (int i = 0; i < NV / 4; ++i) {
      float& v = vs[i & 7];
      v = v * 1.0001f + 0.5f;
      v = v - 0.25f;
      v = __builtin_amdgcn_exp2f(v);
      v = v + vs[(i + 3) & 7] * 0.125f;
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int t = 0; t < a.tiles; ++t) {
    mfma_half(std::integral_constant<int, 0>{});
    exchange();
    bar();
    valu();
    bar();
    mfma_half(std::integral_constant<int, 1>{});
    bar();
    soff += per_tile;
    if (soff + per_tile > a.region) soff = (uint32_t)wave * (uint32_t)NDMA * 1024u;
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) sum += acc[i][0];
#pragma unroll
  for (int i = 0; i < 8; ++i) sum += vs[i];
  if (sum == 12345.678f) a.sink[0] = sum;
}

template <int NM16, int NDMA, int SH0, int SH1, int NV, int XCH>
static double run(const char* name, Args a, int tiles) {
  auto k = probe_share<NM16, NDMA, SH0, SH1, NV, XCH>;
  const int lds = 144 * 1024;
  CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  a.tiles = tiles;
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
  const double flops = (double)NM16 * tiles * 4 * 256 * 16384.0;
  const double tf = flops / (ms * 1e-3) / 1e12;
  printf("SHARE %-78s | %8.3f ms | MFMA %7.1f TFLOP/s (%5.1f%% of 2500) | %.0f ns per tile\n", name, ms, tf, tf / 25.0, ms * 1e6 / tiles);
  fflush(stdout);
  return tf;
}

int main(int argc, char** argv) {
  const uint32_t region = 2u << 20;  // 2 MiB per XCD slice: L2 resident
  char* src;
  CHECK(hipMalloc(&src, (size_t)region * 8));
  CHECK(hipMemset(src, 0x3e, (size_t)region * 8));
  uint32_t* brand;
  CHECK(hipMalloc(&brand, 1024));
  {
    std::vector<uint32_t> h(256);
    uint32_t s = 777u;
    for (auto& x : h) {
      s = s * 1664525u + 1013904223u;
      const uint32_t lo = ((s >> 31) << 15) | (0x3e80 + ((s >> 20) & 0x1ff));
      s = s * 1664525u + 1013904223u;
      const uint32_t hi = ((s >> 31) << 15) | (0x3e80 + ((s >> 20) & 0x1ff));
      x = lo | (hi << 16);
    }
    CHECK(hipMemcpy(brand, h.data(), 1024, hipMemcpyHostToDevice));
  }
  Args a{};
  a.src = src;
  a.region = region;
  a.bsrc = brand;
  CHECK(hipMalloc(&a.sink, 64));
  const int T = argc > 1 ? atoi(argv[1]) : 2048;
  for (int rep = 0; rep < 2; ++rep) {  // twice, interleaved: the second pass is the one to read (clocks settled)
    run<256, 32, 2, 2, 300, 0>("shipped shape: 32-row waves, 2 MFMAs per fragment, 300 VALU", a, T);
    run<256, 32, 4, 4, 600, 16384>("(a) 64-row waves x D/2: 4 per fragment, 600 VALU, 16 KiB partial-S exchange + barrier", a, T);
    run<256, 32, 4, 4, 600, 0>("(a) without the exchange (double softmax only)", a, T);
    run<256, 32, 4, 4, 300, 16384>("(a) without the doubled softmax (exchange only)", a, T);
    run<256, 32, 4, 4, 300, 0>("bound: 4 per fragment in both halves, nothing else changed", a, T);
    run<256, 32, 2, 4, 300, 0>("bound: 4 per fragment in the PV half only (the V^T reads halved, free of charge)", a, T);
  }
  return 0;
}
