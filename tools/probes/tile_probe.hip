// Probe (developer tool): the D = 1024 KV tile of ffpa_fwd_m16_kernel as an instruction skeleton — what do the PLACEMENT of its
// LDS-DMA pieces and the PHASE of its four waves cost?
//
// Per wave and 32-key tile, like the kernel (4 waves = 2 row blocks x 2 D-halves, one wave per SIMD):
//   phase QK : 64 v_mfma_f32_16x16x32_bf16 (32 fragment reads, each feeding two MFMAs) + P0 pieces of V(j), spread evenly
//   barrier A1
//   phase SM : NV VALU instructions (a quarter of them v_exp_f32) in 4 stages, PS pieces of K(j+1) in 4 groups between the stages
//   wait for the V pieces only (vmcnt(PS)), barrier A2
//   phase PV : 64 MFMAs + P1 pieces of K(j+1), spread evenly
//   drain, barrier B
// DEPH > 0: behind every barrier wave w idles w x DEPH x 16 cycles (s_nop 15), so that the four waves of the CU — which leave a
// barrier in the same cycle and whose MFMA streams then tick in lockstep — do not offer their DMA pieces to the one texture
// addresser in the same cycle.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/tile_probe.hip -o tools/probes/bin/tile_probe && tools/probes/bin/tile_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
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
  const char* src;       // stream image, `region` bytes per XCD slice
  uint32_t region;       // bytes of the image one XCD's workgroups walk (wraps)
  int tiles;             // loop iterations
  const uint32_t* bsrc;  // 64 x 4 dwords: the B operand (random bf16)
  float* sink;
  unsigned long long* ticks;  // per wave and phase: s_memtime deltas
};

__device__ __forceinline__ void lds_dma(u32x4 rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

template <int N>
__device__ __forceinline__ void nops16() {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("s_nop 15");
}

// P0 / PS / P1: pieces per wave in the QK / softmax / PV phase (P0 + PS + P1 = 32 at D = 1024); NV: VALU instructions of the softmax
// stand-in; DEPH: see the header; XCH: the partial-S exchange (4 ds_write_b128 + 4 ds_read_b128 per lane around barrier A1)
// Round 5, ROWSPLIT: the "row-split QK^T" layout of the split-D tiles — a wave contracts ALL of D for ONE 16-row half of its row block (its K fragments feed
// one MFMA each: 64 fragment reads in the QK phase instead of 32), runs the softmax of those 16 rows only (NV halves: nothing is computed twice) and hands the
// other D-half's wave its P^T fragments (one ds_write_b128 + one ds_read_b128 per lane around barrier A2) instead of trading fp32 partial S^T tiles around A1.
// ROWSPLIT 2: the same without barrier A1 (no exchange needs it any more; K(j+1) then has to wait for A2: its pieces ride on the PV MFMAs).
template <int P0, int PS, int P1, int NV, int DEPH, bool XCH, bool TIMED, int RD = 1, int ROWSPLIT = 0>
__global__ __launch_bounds__(256) void probe_tile(const Args a) {
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
  const u32x4 braw = *(const u32x4*)(a.bsrc + lane * 4);
  const u32x4 braw2 = *(const u32x4*)(a.bsrc + ((lane + 7) & 63) * 4);
  const bf16x8 b0 = __builtin_bit_cast(bf16x8, braw), b1 = __builtin_bit_cast(bf16x8, braw2);
  constexpr int NACC = 64;
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4)(0.f);
  float vs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) vs[i] = (float)(lane + i) * 1e-3f;
  const uint32_t voff = (uint32_t)lane * 16u;
  constexpr int NP = P0 + PS + P1;
  constexpr uint32_t per_tile = (uint32_t)(4 * NP) * 1024u;
  uint32_t soff = (uint32_t)wave * (uint32_t)NP * 1024u;
  constexpr int PF = 3;
  constexpr int NF = 32;  // fragments per MFMA phase
  bf16x8 fr[4];
  auto frag_read = [&](int f) -> bf16x8 {
    const u32x4 raw = *(LDSAS const u32x4*)(smem + ((f * 1024) & 0xffff) + lane * 16);
    return __builtin_bit_cast(bf16x8, raw);
  };
  auto piece = [&](int p) { lds_dma(rsrc, lds0 + (uint32_t)(((wave * NP + p) * 1024) & (128u * 1024u - 1)), voff, soff + (uint32_t)p * 1024u); };
#pragma unroll
  for (int f = 0; f < PF; ++f) fr[f] = frag_read(f);
  auto mfma_phase = [&](auto hc, auto pc, auto firstc) {
    constexpr int h = decltype(hc)::value, NPH = decltype(pc)::value, first = decltype(firstc)::value;
    constexpr int STEP = NPH > 0 ? NF / NPH : NF;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (RD == 1) fr[(f + PF) & 3] = frag_read((h * NF + f + PF) % (2 * NF));
      bf16x8 fr2 = fr[f & 3];
      if constexpr (ROWSPLIT != 0 && h == 0) fr2 = frag_read((h * NF + f + 17) % (2 * NF));  // the second MFMA of the pair has its own K fragment
      if constexpr (RD == 2) { if (f % 8 == 0) { for (int q = 0; q < 4; ++q) fr[q] = frag_read((h * NF + f + q) % (2 * NF)); } }  // reads in bursts of 4, none in flight at the pieces
      const int ai = (2 * (h * NF + f)) & (NACC - 1);
      acc[ai] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[f & 3], b0, acc[ai], 0, 0, 0);
      if (NPH > 0 && f % STEP == 0 && f / STEP < NPH) piece(first + f / STEP);  // between the pair (the kernel's FFPA_M16_DMA_POS 1)
      acc[ai + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr2, b1, acc[ai + 1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto valu_stage = [&]() {
#pragma unroll
    for (int i = 0; i < NV / 16; ++i) {
      float& v = vs[i & 7];
      v = v * 1.0001f + 0.5f;
      v = v - 0.25f;
      v = __builtin_amdgcn_exp2f(v);
      v = v + vs[(i + 3) & 7] * 0.125f;
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto dephase = [&]() {
    if constexpr (DEPH > 0) {
      if (wave >= 1) nops16<DEPH>();
      if (wave >= 2) nops16<DEPH>();
      if (wave >= 3) nops16<DEPH>();
    }
  };
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long tprev = TIMED ? __builtin_amdgcn_s_memtime() : 0ull;
  auto stamp = [&](int i) {
    if constexpr (TIMED) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      tacc[i] += t - tprev;
      tprev = t;
    }
  };
  LDSAS char* xw = (LDSAS char*)smem + 128 * 1024 + wave * 4096 + lane * 16;
  LDSAS const char* xr = (LDSAS const char*)smem + 128 * 1024 + (wave ^ 1) * 4096 + lane * 16;
  for (int t = 0; t < a.tiles; ++t) {
    dephase();
    mfma_phase(std::integral_constant<int, 0>{}, std::integral_constant<int, P0>{}, std::integral_constant<int, 0>{});
    if constexpr (XCH && ROWSPLIT == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *(LDSAS f32x4*)(xw + i * 1024) = acc[i];
    }
    stamp(0);
    if constexpr (ROWSPLIT != 2) {
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
      __builtin_amdgcn_s_barrier();        // A1
    }
    stamp(1);
    dephase();
    if constexpr (XCH && ROWSPLIT == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 x = *(LDSAS const f32x4*)(xr + i * 1024);
        vs[i] += x[0] + x[3];
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if constexpr (PS >= 4) {
#pragma unroll
        for (int i = 0; i < PS / 4; ++i) piece(P0 + g * (PS / 4) + i);
        __builtin_amdgcn_sched_barrier(0);
      }
      valu_stage();
    }
    if constexpr (ROWSPLIT != 0) {  // this wave's P^T fragments for the other D-half's wave
      const u32x4 pw = {__float_as_uint(vs[0]), __float_as_uint(vs[1]), __float_as_uint(vs[2]), __float_as_uint(vs[3])};
      *(LDSAS u32x4*)xw = pw;
      __builtin_amdgcn_s_waitcnt(0xC07F);
    }
    stamp(2);
    {  // the V pieces (older than the K pieces issued above) have landed: vmcnt(PS)
      constexpr int W = PS > 63 ? 63 : PS;
      __builtin_amdgcn_s_waitcnt(0x0F70 | (W & 15) | ((W >> 4) << 14));
    }
    __builtin_amdgcn_s_barrier();  // A2
    stamp(3);
    dephase();
    if constexpr (ROWSPLIT != 0) {
      const u32x4 pr = *(LDSAS const u32x4*)xr;
      vs[4] += __uint_as_float(pr[0]) + __uint_as_float(pr[3]);
    }
    mfma_phase(std::integral_constant<int, 1>{}, std::integral_constant<int, P1>{}, std::integral_constant<int, P0 + PS>{});
    stamp(4);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();  // B
    stamp(5);
    soff += per_tile;
    if (soff + per_tile > a.region) soff = (uint32_t)wave * (uint32_t)NP * 1024u;
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) sum += acc[i][0];
#pragma unroll
  for (int i = 0; i < 8; ++i) sum += vs[i];
  if (sum == 12345.678f) a.sink[0] = sum;
  if (TIMED && lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) a.ticks[(blockIdx.x * 4 + wave) * 6 + i] = tacc[i];
  }
}

template <int P0, int PS, int P1, int NV, int DEPH, bool XCH = true, int RD = 1, int ROWSPLIT = 0>
static void run_tile(const char* name, Args a, int tiles) {
  const int lds = 144 * 1024;
  a.tiles = tiles;
  float ms = 0;
  {
    auto k = probe_tile<P0, PS, P1, NV, DEPH, XCH, false, RD, ROWSPLIT>;
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
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
  }
  double ph[6] = {0, 0, 0, 0, 0, 0};
  {
    auto k = probe_tile<P0, PS, P1, NV, DEPH, XCH, true, RD, ROWSPLIT>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(k, dim3(256), dim3(256), lds, 0, a);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(256 * 4 * 6);
    CHECK(hipMemcpy(h.data(), a.ticks, h.size() * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < h.size(); ++i) ph[i % 6] += (double)h[i];
    for (int i = 0; i < 6; ++i) ph[i] = ph[i] / (256.0 * 4.0) / tiles * 10.0;  // s_memtime ticks at 100 MHz -> ns per tile
  }
  const double flops = 128.0 * tiles * 4 * 256 * 16384.0;
  printf("TILE %-44s | %8.3f ms %7.1f TFLOP/s (%4.1f%%) %5.0f ns/tile | ns: QK %4.0f  A1 %4.0f  SM %4.0f  A2 %4.0f  PV %4.0f  B %4.0f\n", name, ms,
         flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15 * 100.0, ms * 1e6 / tiles, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5]);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const uint32_t region = 32u << 20;  // K + V of one head at D = 1024, N = 8192: what an XCD's 32 workgroups stream together
  char* src;
  CHECK(hipMalloc(&src, (size_t)region * 8));
  {
    std::vector<uint16_t> h((size_t)region * 8 / 2);
    uint32_t s = 12345u;
    for (auto& x : h) {
      s = s * 1664525u + 1013904223u;
      x = (uint16_t)(((s >> 31) << 15) | ((0x3e80 + ((s >> 20) & 0x1ff))));
    }
    CHECK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  }
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
  CHECK(hipMalloc(&a.ticks, 256 * 4 * 6 * 8));
  const int T = argc > 1 ? atoi(argv[1]) : 4096;
  if (argc > 2 && atoi(argv[2]) == 5) {  // round 5: the row-split QK^T layout against the shipped one
    for (int rep = 0; rep < 3; ++rep) {
      run_tile<16, 16, 0, 160, 0>("shipped layout: 16 / 16 / 0, 160 VALU, S exchange", a, T);
      run_tile<16, 16, 0, 80, 0, true, 1, 1>("row-split: 16 / 16 / 0, 80 VALU, P exchange", a, T);
      run_tile<16, 8, 8, 80, 0, true, 1, 1>("row-split: 16 / 8 / 8", a, T);
      run_tile<16, 0, 16, 80, 0, true, 1, 2>("row-split, no barrier A1: 16 / 0 / 16", a, T);
      run_tile<8, 0, 24, 80, 0, true, 1, 2>("row-split, no barrier A1: 8 / 0 / 24", a, T);
      run_tile<16, 16, 0, 160, 0, true, 1, 1>("row-split reads only (160 VALU kept)", a, T);
      run_tile<16, 16, 0, 80, 0>("shipped layout with 80 VALU (what halving the softmax alone buys)", a, T);
    }
    return 0;
  }
  for (int rep = 0; rep < 2; ++rep) {
    // the shipped placement: V(j) under QK^T, K(j+1) between the softmax stages, nothing under PV
    run_tile<16, 16, 0, 160, 0>("shipped: 16 / 16 / 0", a, T);
    run_tile<16, 16, 0, 160, 1>("shipped, waves 16 cycles apart", a, T);
    run_tile<16, 16, 0, 160, 2>("shipped, waves 32 cycles apart", a, T);
    run_tile<16, 16, 0, 160, 4>("shipped, waves 64 cycles apart", a, T);
    // K(j+1) partly / wholly under the PV MFMAs
    run_tile<16, 8, 8, 160, 0>("16 / 8 / 8", a, T);
    run_tile<16, 8, 8, 160, 1>("16 / 8 / 8, waves 16 cycles apart", a, T);
    run_tile<16, 8, 8, 160, 2>("16 / 8 / 8, waves 32 cycles apart", a, T);
    run_tile<16, 0, 16, 160, 0>("16 / 0 / 16", a, T);
    run_tile<16, 0, 16, 160, 1>("16 / 0 / 16, waves 16 cycles apart", a, T);
    run_tile<16, 0, 16, 160, 2>("16 / 0 / 16, waves 32 cycles apart", a, T);
    // references: no DMA at all; no softmax stand-in; no exchange
    run_tile<0, 0, 0, 160, 0>("no DMA", a, T);
    run_tile<16, 16, 0, 0, 0>("shipped, no VALU", a, T);
    run_tile<16, 16, 0, 160, 0, false>("shipped, no exchange", a, T);
    // do the fragment reads make the pieces expensive?  RD 0: MFMA operands stay in registers (no ds_read at all)
    run_tile<16, 16, 0, 160, 0, true, 0>("shipped, no fragment reads", a, T);
    run_tile<0, 0, 0, 160, 0, true, 0>("no DMA, no fragment reads", a, T);
    run_tile<16, 16, 0, 160, 0, true, 2>("shipped, reads in bursts of 4", a, T);
  }
  return 0;
}
