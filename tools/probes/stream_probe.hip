// Probe (developer tool): what does one gfx950 CU sustain when a one-wave-per-SIMD workgroup mixes the three
// instruction streams of the attention kernel's KV-tile loop —
//   * v_mfma_f32_32x32x16_bf16 (16 rotating accumulators = 256 registers, like O^T),
//   * one ds_read_b128 fragment read per MFMA (the A operand),
//   * 1 KiB LDS-DMA pieces (buffer_load_dwordx4 ... lds) streaming an L2-resident K/V image into LDS,
// with and without the per-tile drain + workgroup barrier?  It answers, with numbers a third party can rerun:
//   (1) the sustained bf16 MFMA rate on random vs zero operands (clock x busy: the DVFS / power ceiling),
//   (2) the L2 -> LDS delivery ceiling of LDS-DMA per CU as a function of the bytes in flight and of the number of waves,
//   (3) what a DMA piece costs a wave that is also issuing MFMAs (the "issue bubble"), at the D = 512 mix
//       (32 pieces per 128 MFMAs) and the D = 1024 mix (32 pieces per 64 MFMAs).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/stream_probe.hip -o tools/probes/bin/stream_probe
//   tools/probes/bin/stream_probe            (prints one PROBE line per configuration)
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define LDSAS __attribute__((address_space(3)))

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

struct Args {
  const char* src;       // stream image, `region` bytes per XCD slice
  uint32_t region;       // bytes of the image one XCD's workgroups walk (wraps)
  int tiles;             // loop iterations
  const uint32_t* bsrc;  // 64 x 4 dwords: the B operand (random or zero bf16)
  float* sink;
  unsigned long long* ticks;  // per wave: s_memtime delta
};

__device__ __forceinline__ void lds_dma(u32x4 rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// NM MFMAs per tile; NDMA 1 KiB pieces per wave per tile, spread evenly over the MFMAs; NREAD: one ds_read_b128 per
// MFMA feeding its A operand; MODE 0: per tile drain (vmcnt(0)) + s_barrier (the kernel's structure, one tile in flight),
// MODE 1: no barrier, wait until only the youngest tile's pieces are in flight (vmcnt(NDMA)), MODE 2: never wait
// inside the loop (vmcnt saturates at 63: the deepest queue the hardware keeps).
// SWZ: per-lane source offset of a piece: 0 = lane-linear, 1 = 16-byte slots XOR-permuted inside 256-byte groups (the kernel's K image),
// 2 = 64-byte quarters permuted inside 256-byte groups (the kernel's V image)
template <int NM, int NDMA, int NREAD, int MODE, bool MFMA_ON, int SWZ = 0, int NACC = 16>
__global__ __launch_bounds__(MFMA_ON ? (NACC == 16 ? 256 : 512) : 1024) void probe(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwave = blockDim.x >> 6;
  const int xcd = blockIdx.x & 7;
  // every workgroup of an XCD walks the same image in lockstep (like row tiles of one head sharing K/V through L2)
  const char* base = a.src + (size_t)xcd * a.region;
  const uint64_t ba = (uint64_t)base;
  const u32x4 rsrc = {(uint32_t)ba, (uint32_t)(ba >> 32) & 0xffffu, a.region, 0x00020000u};
  const uint32_t lds0 = (uint32_t)(uintptr_t)(LDSAS char*)smem;
  const uint32_t per_tile = (uint32_t)(nwave * (NDMA > 0 ? NDMA : 1) * 1024);  // bytes of the image per tile
  const uint32_t lds_ring = 128u * 1024u;
  // the fragment-read region: first 64 KiB of LDS, lane-linear (conflict-free b128)
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((LDSAS uint32_t*)smem)[i] = 0x3f803f80u ^ (uint32_t)(i * 2654435761u >> 12 & 0x00ff00ffu);
  __syncthreads();
  const u32x4 braw = *(const u32x4*)(a.bsrc + lane * 4);
  const bf16x8 bfrag = __builtin_bit_cast(bf16x8, braw);
  f32x16 acc[NACC];  // 16 = a one-wave-per-SIMD kernel's 256 accumulator registers; 8 = two waves per SIMD
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x16)(0.f);
  const uint32_t voff = (uint32_t)lane * 16u;
  uint32_t soff = (uint32_t)wave * (uint32_t)(NDMA > 0 ? NDMA : 1) * 1024u;
  // fragment reads run PF MFMAs ahead of their consumer through a ring of 8 registers sets (the kernel's software pipeline)
  constexpr int PF = 6;
  static_assert(NM % 8 == 0, "ring indices must be static across tiles");
  bf16x8 fr[8];
  auto frag_read = [&](int n) -> bf16x8 {
    const u32x4 raw = *(LDSAS const u32x4*)(smem + ((n * 1024) & 0xffff) + lane * 16);
    return __builtin_bit_cast(bf16x8, raw);
  };
#pragma unroll
  for (int i = 0; i < 8; ++i) fr[i] = bfrag;
  if constexpr (NREAD > 0) {
#pragma unroll
    for (int n = 0; n < PF; ++n) fr[n] = frag_read(n);
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < a.tiles; ++t) {
    constexpr int STEP = NDMA > 0 ? (NM / NDMA > 0 ? NM / NDMA : 1) : 1;
#pragma unroll
    for (int n = 0; n < NM; ++n) {
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NREAD > 0) fr[(n + PF) & 7] = frag_read((n + PF) % NM);
      if constexpr (NDMA > 0) {
        if (n % STEP == 0 && n / STEP < NDMA) {
          const int piece = n / STEP;
          // destination: this wave's slot in a ring over LDS (the fragment reads only need *some* data there)
          const uint32_t dst = lds0 + (uint32_t)(((wave * NDMA + piece) * 1024) & (lds_ring - 1));
          const uint32_t vo = SWZ == 1 ? (uint32_t)((lane ^ (piece & 15)) << 4) : SWZ == 2 ? (uint32_t)((lane ^ ((piece & 3) << 2)) << 4) : voff;
          lds_dma(rsrc, dst, vo, soff + (uint32_t)piece * 1024u);
        }
      }
      if constexpr (MFMA_ON) acc[n % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[n & 7], bfrag, acc[n % NACC], 0, 0, 0);
      else if constexpr (NREAD > 0) asm volatile("" ::"v"(fr[n & 7]));
    }
    if constexpr (NM == 0 && NDMA > 0) {}
    __builtin_amdgcn_sched_barrier(0);
    soff += per_tile;
    if (soff + per_tile > a.region) soff = (uint32_t)wave * (uint32_t)(NDMA > 0 ? NDMA : 1) * 1024u;
    if constexpr (MODE == 0) {
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __builtin_amdgcn_s_barrier();
    } else if constexpr (MODE == 1 && NDMA > 0) {
      constexpr int W = NDMA > 63 ? 63 : NDMA;
      __builtin_amdgcn_s_waitcnt(0x0F70 | (W & 15) | ((W >> 4) << 14));
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][7];
  if (s == 12345.678f) a.sink[0] = s;
  if (lane == 0) a.ticks[blockIdx.x * nwave + wave] = t1 - t0;
}

// bare v_mfma_f32_16x16x32_bf16 stream (64 rotating 4-register accumulators = the same 256 registers): does the other bf16
// shape sustain a different clock / rate on random operands?
typedef __attribute__((ext_vector_type(4))) float f32x4p;
__global__ __launch_bounds__(256) void probe_mfma16(const Args a) {
  const int lane = threadIdx.x & 63;
  const u32x4 braw = *(const u32x4*)(a.bsrc + lane * 4);
  const bf16x8 bfrag = __builtin_bit_cast(bf16x8, braw);
  f32x4p acc[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = (f32x4p)(0.f);
  for (int t = 0; t < a.tiles; ++t) {
#pragma unroll
    for (int n = 0; n < 128; ++n) {
      __builtin_amdgcn_sched_barrier(0);
      acc[n & 63] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfrag, bfrag, acc[n & 63], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) s += acc[i][0];
  if (s == 12345.678f) a.sink[0] = s;
}

// The kernel's instruction mix rebuilt on the 16x16x32 shape: one ds_read_b128 A fragment feeds TWO MFMAs (the two 16-row halves
// of a wave's 32 query rows are its B operands), so LDS bytes per FLOP equal the 32x32x16 mix; NM16 MFMAs of 16 K FLOP, NDMA
// pieces and NM16 / 2 fragment reads per wave and tile.
template <int NM16, int NDMA, int MODE, int SHARE = 2, int ORDER = 0>
__global__ __launch_bounds__(256) void probe_mix16(const Args a) {
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
  f32x4p acc[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = (f32x4p)(0.f);
  const uint32_t voff = (uint32_t)lane * 16u;
  uint32_t soff = (uint32_t)wave * (uint32_t)(NDMA > 0 ? NDMA : 1) * 1024u;
  const uint32_t per_tile = 4u * (uint32_t)(NDMA > 0 ? NDMA : 1) * 1024u;
  constexpr int PF = 3;  // fragments ahead (each feeds two MFMAs)
  constexpr int NF = NM16 / SHARE;  // SHARE = 2: one fragment feeds two MFMAs (a wave's two 16-row halves); 1: a 16-row wave, one MFMA per fragment
  bf16x8 fr[4];
  auto frag_read = [&](int f) -> bf16x8 {
    const u32x4 raw = *(LDSAS const u32x4*)(smem + ((f * 1024) & 0xffff) + lane * 16);
    return __builtin_bit_cast(bf16x8, raw);
  };
#pragma unroll
  for (int f = 0; f < PF; ++f) fr[f] = frag_read(f);
  for (int t = 0; t < a.tiles; ++t) {
    constexpr int STEP = NDMA > 0 ? NF / NDMA : 1;
    // ORDER: where the DMA piece sits relative to the fragment's two MFMAs: 0 = in front, 1 = between, 2 = behind.  (Giving every wave
    // its own slot between two pieces was tried with one unrolled loop body per wave: 749 TFLOPS — four bodies thrash the instruction
    // cache; with a wave-uniform branch per fragment instead: 1082.)
    auto tile_body = [&](auto slotc) {
      constexpr int SLOT = decltype(slotc)::value;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        __builtin_amdgcn_sched_barrier(0);
        fr[(f + PF) & 3] = frag_read((f + PF) % NF);
        auto dma = [&]() {
          if constexpr (NDMA > 0) {
            if (f % STEP == SLOT % STEP && f / STEP < NDMA) {
              const int piece = f / STEP;
              lds_dma(rsrc, lds0 + (uint32_t)(((wave * NDMA + piece) * 1024) & (128u * 1024u - 1)), voff, soff + (uint32_t)piece * 1024u);
            }
          }
        };
        if constexpr (ORDER == 0) dma();
        if constexpr (SHARE == 2) {
          acc[(2 * f) & 63] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[f & 3], b0, acc[(2 * f) & 63], 0, 0, 0);
          if constexpr (ORDER == 1) {
            __builtin_amdgcn_sched_barrier(0);
            dma();
            __builtin_amdgcn_sched_barrier(0);
          }
          acc[(2 * f + 1) & 63] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[f & 3], b1, acc[(2 * f + 1) & 63], 0, 0, 0);
        } else {
          acc[f & 63] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[f & 3], (f & 1) ? b1 : b0, acc[f & 63], 0, 0, 0);
        }
        if constexpr (ORDER == 2) {
          __builtin_amdgcn_sched_barrier(0);
          dma();
        }
      }
    };
    tile_body(std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    soff += per_tile;
    if (soff + per_tile > a.region) soff = (uint32_t)wave * (uint32_t)(NDMA > 0 ? NDMA : 1) * 1024u;
    if constexpr (MODE == 0) {
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __builtin_amdgcn_s_barrier();
    } else if constexpr (NDMA > 0) {
      constexpr int W = NDMA > 63 ? 63 : NDMA;
      __builtin_amdgcn_s_waitcnt(0x0F70 | (W & 15) | ((W >> 4) << 14));
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) s += acc[i][0];
  if (s == 12345.678f) a.sink[0] = s;
}

// One wave per SIMD vs two: the D = 512 tile WITH its softmax phase.  Per tile and wave: NM16 MFMAs in two halves (QK^T, PV) with
// their fragment reads (SHARE MFMAs per read) and NDMA pieces, NV VALU instructions (a quarter of them v_exp_f32) between the halves,
// three barriers.  WAVES = 4: 32-row waves (NM16 = 256, SHARE = 2, NV = 300); WAVES = 8: 16-row waves, two per SIMD, each half the
// work (NM16 = 128, SHARE = 1, NV = 150, 128 accumulator registers) — the second wave of a SIMD can run MFMAs under the first one's VALU.
template <int WAVES, int NM16, int NDMA, int SHARE, int NV, bool PINGPONG = false>
__global__ __launch_bounds__(WAVES * 64) void probe_soft(const Args a) {
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
  constexpr int NACC = WAVES == 4 ? 64 : 32;
  f32x4p acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4p)(0.f);
  float vs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) vs[i] = (float)(lane + i) * 1e-3f;
  const uint32_t voff = (uint32_t)lane * 16u;
  constexpr uint32_t per_tile = (uint32_t)(WAVES * NDMA) * 1024u;
  uint32_t soff = (uint32_t)wave * (uint32_t)NDMA * 1024u;
  constexpr int PF = 3;
  constexpr int NF = NM16 / SHARE;
  constexpr int STEP = NF / NDMA;
  bf16x8 fr[4];
  auto frag_read = [&](int f) -> bf16x8 {
    const u32x4 raw = *(LDSAS const u32x4*)(smem + ((f * 1024) & 0xffff) + lane * 16);
    return __builtin_bit_cast(bf16x8, raw);
  };
#pragma unroll
  for (int f = 0; f < PF; ++f) fr[f] = frag_read(f);
  auto mfma_half = [&](auto hc) {
    constexpr int h = decltype(hc)::value;
#pragma unroll
    for (int f = h * (NF / 2); f < (h + 1) * (NF / 2); ++f) {
      __builtin_amdgcn_sched_barrier(0);
      fr[(f + PF) & 3] = frag_read((f + PF) % NF);
      if (f % STEP == 0 && f / STEP < NDMA)
        lds_dma(rsrc, lds0 + (uint32_t)(((wave * NDMA + f / STEP) * 1024) & (128u * 1024u - 1)), voff, soff + (uint32_t)(f / STEP) * 1024u);
      if constexpr (SHARE == 2) {
        acc[(2 * f) & (NACC - 1)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[f & 3], b0, acc[(2 * f) & (NACC - 1)], 0, 0, 0);
        acc[(2 * f + 1) & (NACC - 1)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[f & 3], b1, acc[(2 * f + 1) & (NACC - 1)], 0, 0, 0);
      } else {
        acc[f & (NACC - 1)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[f & 3], (f & 1) ? b1 : b0, acc[f & (NACC - 1)], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto valu = [&]() {
    // the softmax stand-in: NV VALU instructions, 8 independent chains, every fourth a quarter-rate exponential
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
  auto bar = [&]() {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
  };
  // PINGPONG: the second wave of every SIMD (waves 4 .. 7) runs one phase behind the first: while one of the two is in its softmax
  // stand-in the other issues MFMAs (phases are still separated by workgroup barriers, three per tile)
  const bool late = PINGPONG && wave >= WAVES / 2;
  if (late) {  // one MFMA phase ahead of the common loop: from here on this wave is one barrier-to-barrier slot behind its SIMD partner
    mfma_half(std::integral_constant<int, 1>{});
    bar();
  }
  for (int t = 0; t < a.tiles; ++t) {
    mfma_half(std::integral_constant<int, 0>{});
    bar();
    valu();
    bar();
    mfma_half(std::integral_constant<int, 1>{});
    bar();
    soff += per_tile;
    if (soff + per_tile > a.region) soff = (uint32_t)wave * (uint32_t)NDMA * 1024u;
  }
  if (PINGPONG && !late) bar();  // (the early waves meet the late waves' last barrier)
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) sum += acc[i][0];
#pragma unroll
  for (int i = 0; i < 8; ++i) sum += vs[i];
  if (sum == 12345.678f) a.sink[0] = sum;
}

template <int WAVES, int NM16, int NDMA, int SHARE, int NV, bool PINGPONG = false>
static void run_soft(const char* name, Args a, const uint32_t* brand, int tiles) {
  auto k = probe_soft<WAVES, NM16, NDMA, SHARE, NV, PINGPONG>;
  const int lds = 144 * 1024;
  CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  a.tiles = tiles;
  a.bsrc = brand;
  hipLaunchKernelGGL(k, dim3(256), dim3(WAVES * 64), lds, 0, a);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(WAVES * 64), lds, 0, a);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 3;
  const double flops = (double)NM16 * tiles * WAVES * 256 * 16384.0;
  printf("PROBE %-52s waves/CU %2d tiles %5d | %8.3f ms | MFMA %7.1f TFLOP/s (%5.1f%% of 2500) | %.0f ns per tile\n", name, WAVES, tiles, ms,
         flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15 * 100.0, ms * 1e6 / tiles);
  fflush(stdout);
}

// Loader-wave layout (D = 1024 candidate): waves 0 .. 2 are 16-row MFMA waves (NM16 MFMAs per tile, one fragment read per MFMA, HELP DMA
// pieces each, spread over their MFMAs), wave 3 only issues LDS-DMA (NLOAD pieces per tile in two bursts); three workgroup barriers
// per tile like the kernel (the loader drains its queue before the second and third).  FLOPs counted for the three MFMA waves.
template <int NM16, int NLOAD, int HELP>
__global__ __launch_bounds__(256) void probe_loader(const Args a) {
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
  const uint32_t voff = (uint32_t)lane * 16u;
  constexpr uint32_t per_tile = (uint32_t)(NLOAD + 3 * HELP) * 1024u;
  uint32_t soff = 0;
  if (wave == 3) {
    for (int t = 0; t < a.tiles; ++t) {
#pragma unroll
      for (int i = 0; i < NLOAD / 2; ++i) lds_dma(rsrc, lds0 + (uint32_t)((i * 1024) & (128u * 1024u - 1)), voff, soff + (uint32_t)i * 1024u);
      __builtin_amdgcn_s_barrier();  // A1
#pragma unroll
      for (int i = NLOAD / 2; i < NLOAD; ++i) lds_dma(rsrc, lds0 + (uint32_t)((i * 1024) & (128u * 1024u - 1)), voff, soff + (uint32_t)i * 1024u);
      __builtin_amdgcn_s_waitcnt(0x0F70 | ((NLOAD / 2) & 15) | (((NLOAD / 2) >> 4) << 14));  // the first burst has landed
      __builtin_amdgcn_s_barrier();  // A2
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __builtin_amdgcn_s_barrier();  // B
      soff += per_tile;
      if (soff + per_tile > a.region) soff = 0;
    }
    return;
  }
  const u32x4 braw = *(const u32x4*)(a.bsrc + lane * 4);
  const bf16x8 b0 = __builtin_bit_cast(bf16x8, braw);
  f32x4p acc[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = (f32x4p)(0.f);
  constexpr int PF = 3;
  bf16x8 fr[4];
  auto frag_read = [&](int f) -> bf16x8 {
    const u32x4 raw = *(LDSAS const u32x4*)(smem + ((f * 1024) & 0xffff) + lane * 16);
    return __builtin_bit_cast(bf16x8, raw);
  };
#pragma unroll
  for (int f = 0; f < PF; ++f) fr[f] = frag_read(f);
  constexpr int HALF = NM16 / 2;
  constexpr int STEP = HELP > 0 ? NM16 / HELP : 1;
  for (int t = 0; t < a.tiles; ++t) {
#pragma unroll
    for (int f = 0; f < NM16; ++f) {
      __builtin_amdgcn_sched_barrier(0);
      fr[(f + PF) & 3] = frag_read((f + PF) % NM16);
      if constexpr (HELP > 0) {
        if (f % STEP == 0 && f / STEP < HELP)
          lds_dma(rsrc, lds0 + (uint32_t)(((NLOAD + wave * HELP + f / STEP) * 1024) & (128u * 1024u - 1)), voff, soff + (uint32_t)(NLOAD + wave * HELP + f / STEP) * 1024u);
      }
      acc[f & 63] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[f & 3], b0, acc[f & 63], 0, 0, 0);
      if (f == HALF - 1) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();  // A1 (after the first GEMM)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();  // A2 (before the second)
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();  // B
    soff += per_tile;
    if (soff + per_tile > a.region) soff = 0;
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) sum += acc[i][0];
  if (sum == 12345.678f) a.sink[0] = sum;
}

template <int NM16, int NLOAD, int HELP>
static void run_loader(const char* name, Args a, const uint32_t* brand, int tiles) {
  auto k = probe_loader<NM16, NLOAD, HELP>;
  const int lds = 144 * 1024;
  CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  a.tiles = tiles;
  a.bsrc = brand;
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
  const double flops = (double)NM16 * tiles * 3 * 256 * 16384.0;
  const double bytes_cu = (double)(NLOAD + 3 * HELP) * 1024.0 * tiles;
  printf("PROBE %-46s 3 mfma waves + loader tiles %5d | %8.3f ms | MFMA %7.1f TFLOP/s (%5.1f%% of 2500) | LDS-DMA %6.2f TB/s chip | %.0f ns per tile\n", name, tiles,
         ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15 * 100.0, bytes_cu * 256 / (ms * 1e-3) / 1e12, ms * 1e6 / tiles);
  fflush(stdout);
}

template <int NM16, int NDMA, int MODE, int SHARE = 2, int ORDER = 0>
static void run16(const char* name, Args a, const uint32_t* brand, int tiles) {
  auto k = probe_mix16<NM16, NDMA, MODE, SHARE, ORDER>;
  const int lds = 144 * 1024;
  CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  a.tiles = tiles;
  a.bsrc = brand;
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
  const double bytes_cu = (double)NDMA * 1024.0 * 4 * tiles;
  printf("PROBE %-34s waves/CU  4 ops=random tiles %5d | %8.3f ms | MFMA %7.1f TFLOP/s (%5.1f%% of 2500) | LDS-DMA %6.2f TB/s chip\n", name, tiles, ms,
         flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15 * 100.0, bytes_cu * 256 / (ms * 1e-3) / 1e12);
  fflush(stdout);
}

static double g_clock_hint = 0;

template <int NM, int NDMA, int NREAD, int MODE, bool MFMA_ON, int SWZ = 0, int NACC = 16>
static void run(const char* name, Args a, int threads, const uint32_t* brand, const uint32_t* bzero, bool zero_ops, int tiles) {
  auto k = probe<NM, NDMA, NREAD, MODE, MFMA_ON, SWZ, NACC>;
  const int lds = 144 * 1024;
  CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  a.tiles = tiles;
  a.bsrc = zero_ops ? bzero : brand;
  const int grid = 256;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, 0, a);  // warm-up (also warms L2 / MALL)
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  const int reps = 3;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, 0, a);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const int nw = threads / 64;
  std::vector<unsigned long long> ticks(grid * nw);
  CHECK(hipMemcpy(ticks.data(), a.ticks, ticks.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long tmax = 0;
  double tsum = 0;
  for (auto t : ticks) {
    tmax = t > tmax ? t : tmax;
    tsum += (double)t;
  }
  const double tavg = tsum / ticks.size();
  // __builtin_readcyclecounter = s_memtime: a constant-rate counter (100 MHz on gfx950); report it and derive nothing from it
  const double mfma_per_simd = (double)NM * tiles * (nw / 4.0);
  const double flops = MFMA_ON ? mfma_per_simd * 4 * 256 * 32768.0 : 0;
  const double bytes_cu = (double)NDMA * 1024.0 * nw * tiles;
  printf("PROBE %-34s waves/CU %2d ops=%s tiles %5d | %8.3f ms | MFMA %7.1f TFLOP/s (%5.1f%% of 2500) | LDS-DMA %6.2f TB/s chip = %6.1f GB/s/CU"
         " | ns per tile %7.1f | memtime ticks avg %.0f max %llu\n",
         name, nw, zero_ops ? "zero  " : "random", tiles, ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15 * 100.0,
         bytes_cu * 256 / (ms * 1e-3) / 1e12, bytes_cu / (ms * 1e-3) / 1e9, ms * 1e6 / tiles, tavg, tmax);
  fflush(stdout);
  (void)g_clock_hint;
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
      // random bf16 in roughly N(0,1) range: sign + exponent 0x3f/0x3e/0x40 + random mantissa
      x = (uint16_t)(((s >> 31) << 15) | ((0x3e80 + ((s >> 20) & 0x1ff))));
    }
    CHECK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  }
  uint32_t *brand, *bzero;
  CHECK(hipMalloc(&brand, 1024));
  CHECK(hipMalloc(&bzero, 1024));
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
    CHECK(hipMemset(bzero, 0, 1024));
  }
  Args a{};
  a.src = src;
  a.region = region;
  CHECK(hipMalloc(&a.sink, 64));
  CHECK(hipMalloc(&a.ticks, 256 * 16 * 8));
  const int T = argc > 1 ? atoi(argv[1]) : 4096;

  // (1) bare MFMA stream: the sustained-clock x issue ceiling, random vs zero operands
  run<64, 0, 0, 2, true>("mfma_only", a, 256, brand, bzero, false, T * 2);
  run<64, 0, 0, 2, true>("mfma_only", a, 256, brand, bzero, true, T * 2);
  // (1b) + one conflict-free ds_read_b128 per MFMA (A operand from LDS)
  run<64, 0, 1, 2, true>("mfma+ldsread", a, 256, brand, bzero, false, T * 2);
  // (2) LDS-DMA alone: delivery ceiling vs queue depth and waves per CU
  run<32, 32, 0, 0, false>("dma_only drain+barrier/tile", a, 256, brand, bzero, false, T);
  run<32, 32, 0, 1, false>("dma_only 32..64 in flight", a, 256, brand, bzero, false, T);
  run<32, 32, 0, 2, false>("dma_only queue saturated", a, 256, brand, bzero, false, T);
  run<32, 32, 0, 2, false>("dma_only queue saturated", a, 512, brand, bzero, false, T / 2);
  run<32, 32, 0, 2, false>("dma_only queue saturated", a, 1024, brand, bzero, false, T / 4);
  run<16, 16, 0, 1, false>("dma_only 16..32 in flight", a, 256, brand, bzero, false, T * 2);
  // (2b) fragment reads alone and DMA + fragment reads (LDS port sharing), no MFMA
  run<64, 0, 1, 2, false>("ldsread_only", a, 256, brand, bzero, false, T * 2);
  run<64, 32, 1, 1, false>("dma32+ldsread64", a, 256, brand, bzero, false, T);
  // (3) the kernel's mixes.  D = 1024 tile: 64 MFMAs + 32 pieces per wave; D = 512 tile: 128 MFMAs + 32 pieces
  run<64, 32, 0, 1, true>("D1024mix mfma64+dma32", a, 256, brand, bzero, false, T);
  run<64, 32, 1, 1, true>("D1024mix +ldsread", a, 256, brand, bzero, false, T);
  run<64, 32, 1, 0, true>("D1024mix +ldsread +barrier", a, 256, brand, bzero, false, T);
  run<64, 16, 1, 1, true>("half the DMA: mfma64+dma16+read", a, 256, brand, bzero, false, T);
  run<128, 32, 0, 1, true>("D512mix mfma128+dma32", a, 256, brand, bzero, false, T / 2);
  run<128, 32, 1, 1, true>("D512mix +ldsread", a, 256, brand, bzero, false, T / 2);
  run<128, 32, 1, 0, true>("D512mix +ldsread +barrier", a, 256, brand, bzero, false, T / 2);
  // (4) does the bank-conflict swizzle of the SOURCE offsets (lanes of a piece no longer ascend through memory) cost delivery rate?
  run<32, 32, 0, 1, false, 1>("dma_only, 16-B slot swizzle (K)", a, 256, brand, bzero, false, T);
  run<32, 32, 0, 1, false, 2>("dma_only, 64-B quarter swizzle (V)", a, 256, brand, bzero, false, T);
  run<64, 32, 1, 1, true, 1>("D1024mix +ldsread, K swizzle", a, 256, brand, bzero, false, T);
  run<64, 32, 1, 1, true, 2>("D1024mix +ldsread, V swizzle", a, 256, brand, bzero, false, T);
  {  // (1c) the 16x16x32 shape: 128 MFMAs of 16 K FLOP per "tile" = the FLOPs of 64 32x32x16 MFMAs
    for (int zero = 0; zero < 2; ++zero) {
      Args b = a;
      b.tiles = T * 2;
      b.bsrc = zero ? bzero : brand;
      hipLaunchKernelGGL(probe_mfma16, dim3(256), dim3(256), 0, 0, b);
      CHECK(hipDeviceSynchronize());
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0));
      CHECK(hipEventCreate(&e1));
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(probe_mfma16, dim3(256), dim3(256), 0, 0, b);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      ms /= 3;
      const double flops = 128.0 * b.tiles * 4 * 256 * 16384.0;
      printf("PROBE %-34s waves/CU  4 ops=%s tiles %5d | %8.3f ms | MFMA %7.1f TFLOP/s (%5.1f%% of 2500)\n", "mfma_only 16x16x32", zero ? "zero  " : "random", b.tiles, ms,
             flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15 * 100.0);
    }
  }
  // (1d) the kernel's mixes on the 16x16x32 shape (A fragment shared by two MFMAs)
  run16<256, 0, 1>("16x16x32: mfma + shared-A reads", a, brand, T);
  run16<256, 32, 1>("16x16x32: D512mix (256 mfma16+32 dma)", a, brand, T / 2);
  run16<256, 32, 0>("16x16x32: D512mix +barrier", a, brand, T / 2);
  run16<128, 32, 1>("16x16x32: D1024mix (128 mfma16+32 dma)", a, brand, T);
  run16<128, 32, 0>("16x16x32: D1024mix +barrier", a, brand, T);
  // a 16-row wave (D = 1024 unsplit: every fragment feeds ONE MFMA — twice the LDS reads per FLOP, no partial-S exchange)
  run16<256, 0, 1, 1>("16x16x32: mfma + one read per mfma", a, brand, T);
  run16<128, 32, 0, 1>("16x16x32: D1024mix, read per mfma +barrier", a, brand, T);
  // the D = 512 tile with its softmax phase: one 32-row wave per SIMD vs two 16-row waves per SIMD
  run_soft<4, 256, 32, 2, 0>("D512 tile, 4 waves, no softmax stand-in", a, brand, T / 2);
  run_soft<4, 256, 32, 2, 300>("D512 tile, 4 waves x 32 rows, 300 VALU", a, brand, T / 2);
  run_soft<8, 128, 16, 1, 0>("D512 tile, 8 waves x 16 rows, no softmax stand-in", a, brand, T / 2);
  run_soft<8, 128, 16, 1, 150>("D512 tile, 8 waves x 16 rows, 150 VALU each", a, brand, T / 2);
  run_soft<8, 128, 16, 1, 150, true>("D512 tile, 8 waves x 16 rows, 150 VALU, ping-pong", a, brand, T / 2);
  run_soft<8, 128, 16, 1, 0, true>("D512 tile, 8 waves x 16 rows, no VALU, ping-pong", a, brand, T / 2);
  // D = 1024 with a loader wave: 3 x 16-row MFMA waves (128 MFMA16 per 32-key tile each), 128 pieces per tile
  run_loader<128, 128, 0>("loader: 128 mfma16 x3, loader issues all 128", a, brand, T);
  run_loader<128, 104, 8>("loader: 128 mfma16 x3, loader 104 + 8 per mfma wave", a, brand, T);
  run_loader<128, 80, 16>("loader: 128 mfma16 x3, loader 80 + 16 per mfma wave", a, brand, T);
  run_loader<128, 0, 0>("loader: 128 mfma16 x3, no dma at all", a, brand, T);
  // what a DMA piece costs the 16x16x32 stream, and whether its place among the MFMAs matters
  run16<256, 16, 1>("16x16x32: 256 mfma16 + 16 dma", a, brand, T / 2);
  run16<256, 64, 1>("16x16x32: 256 mfma16 + 64 dma", a, brand, T / 2);
  run16<256, 32, 1, 2, 1>("16x16x32: D512mix, dma between the pair", a, brand, T / 2);
  run16<256, 32, 1, 2, 2>("16x16x32: D512mix, dma behind the pair", a, brand, T / 2);
  // (5) would TWO waves per SIMD (8 per CU, 256 registers each: 128 accumulators) hide the in-order stalls?  Same work per CU and
  // tile as the D = 512 / D = 1024 mixes, split over 8 waves (half the MFMAs, pieces and reads per wave)
  run<64, 0, 0, 2, true, 0, 8>("8 waves: mfma_only", a, 512, brand, bzero, false, T * 2);
  run<64, 16, 1, 1, true, 0, 8>("8 waves: D512mix (64 mfma + 16 dma + reads)/wave", a, 512, brand, bzero, false, T / 2 * 2);
  run<64, 16, 1, 0, true, 0, 8>("8 waves: D512mix +barrier", a, 512, brand, bzero, false, T / 2 * 2);
  run<32, 16, 1, 1, true, 0, 8>("8 waves: D1024mix (32 mfma + 16 dma + reads)/wave", a, 512, brand, bzero, false, T * 2);
  run<32, 16, 1, 0, true, 0, 8>("8 waves: D1024mix +barrier", a, 512, brand, bzero, false, T * 2);
  return 0;
}
