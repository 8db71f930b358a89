// operand_order_probe.hip — does the ORDER in which a wave presents operands to the matrix core change the power-capped MFMA rate?
//
// The prefill kernels run against the chip's power cap (profiles/NOTES.md section 3: bare v_mfma_f32_16x16x32_bf16 on random operands sustains
// 0.80 - 0.85 of the 2.4 GHz peak, on zeros 0.99): energy per MFMA is what the clock pays for, and the energy of a multiplier array depends on how
// many operand bits toggle between consecutive instructions.  The kernel's loops keep ONE operand fixed over 2 (3 in the wide tile) consecutive MFMAs
// (the K / V^T fragment, operand A, shared by the row halves).  This probe issues the same number of MFMAs on the same random registers in different
// orders: A held for 1 / 2 / 4 / 16 instructions, B held for 2 / 4 / 16, both held.  One wave per SIMD, 4 waves per CU, every CU, ~ 50 ms per arm.
//
//   hipcc --offload-arch=gfx950 -O2 tools/probes/operand_order_probe.hip -o tools/probes/bin/operand_order_probe && tools/probes/bin/operand_order_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 v8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int NA = 16, NB = 16, NACC = 16;

// MODE: which (a, b) pair MFMA number n of a 256-instruction block uses
//   0: a = n % 16, b = n % 16 (+ n / 16 rotation)   — both operands change every instruction
//   1: A held for 2   (the 32-row tile: one K fragment, two row halves)        a = (n / 2) % 16, b = n % 16
//   2: A held for 4                                                            a = (n / 4) % 16, b = n % 16
//   3: A held for 16                                                           a = (n / 16) % 16, b = n % 16
//   4: B held for 2                                                            a = n % 16, b = (n / 2) % 16
//   5: B held for 4
//   6: B held for 16
//   7: both held for 16 (different accumulators only)
template <int MODE>
__global__ __launch_bounds__(256) void probe(const u32x4* __restrict__ src, float* __restrict__ out, int iters) {
  v8 A[NA], B[NB];
  const int lane = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NA; ++i) A[i] = __builtin_bit_cast(v8, src[(i * 256 + lane) & 4095]);
#pragma unroll
  for (int i = 0; i < NB; ++i) B[i] = __builtin_bit_cast(v8, src[((i + 16) * 256 + lane) & 4095]);
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4)(0.f);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < 256; ++n) {
      constexpr int dummy = 0;
      (void)dummy;
      const int a = MODE == 0 ? n % 16 : MODE == 1 ? (n / 2) % 16 : MODE == 2 ? (n / 4) % 16 : MODE == 3 ? (n / 16) % 16 : MODE == 7 ? (n / 16) % 16 : n % 16;
      const int b = MODE == 0 ? (n + n / 16) % 16 : MODE <= 3 ? n % 16 : MODE == 4 ? (n / 2) % 16 : MODE == 5 ? (n / 4) % 16 : (n / 16) % 16;
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[n % NACC]) : "v"(A[a]), "v"(B[b]));
    }
  }
  asm volatile("s_nop 15\n\ts_nop 3");
  f32x4 s = (f32x4)(0.f);
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  if (s[0] == 12345.678f) out[blockIdx.x] = s[1] + s[2] + s[3];
}

template <int MODE>
int run(const char* name, const u32x4* src, float* out, int cus) {
  const int iters = 40000, grid = cus * 1;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {  // (the first launch warms the clocks)
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, src, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)grid * 4.0 * iters * 256.0 * 16.0 * 16.0 * 32.0 * 2.0;
  printf("ORDER %-46s %8.3f ms  %7.1f TFLOP/s (%.3f of 2500)\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 2500.0);
  return 0;
}

int main() {
  int cus = 256;
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  std::vector<uint32_t> h(4096 * 4);
  uint32_t s = 4242u;
  for (auto& x : h) {
    s = s * 1664525u + 1013904223u;
    const uint32_t lo = ((s >> 31) << 15) | (0x3e80 + ((s >> 20) & 0x1ff));
    s = s * 1664525u + 1013904223u;
    const uint32_t hi = ((s >> 31) << 15) | (0x3e80 + ((s >> 20) & 0x1ff));
    x = lo | (hi << 16);
  }
  u32x4* src;
  float* out;
  CHECK(hipMalloc(&src, h.size() * 4));
  CHECK(hipMalloc(&out, 4096 * 4));
  CHECK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  for (int round = 0; round < 2; ++round) {  // (two rounds: drift of the box shows as a difference between them)
    if (run<0>("both operands change every MFMA", src, out, cus)) return 1;
    if (run<1>("A held for 2 (the 32-row tile's loops)", src, out, cus)) return 1;
    if (run<2>("A held for 4", src, out, cus)) return 1;
    if (run<3>("A held for 16", src, out, cus)) return 1;
    if (run<4>("B held for 2", src, out, cus)) return 1;
    if (run<5>("B held for 4", src, out, cus)) return 1;
    if (run<6>("B held for 16", src, out, cus)) return 1;
    if (run<7>("A and B held for 16", src, out, cus)) return 1;
  }
  return 0;
}
