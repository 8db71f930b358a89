// gap_probe.hip — how many single-issue instructions does ONE wave per SIMD hide in the gap of its own v_mfma_f32_16x16x32_bf16 (16 cycles)?
//
// The prefill kernels run one wave per SIMD (the wave owns the SIMD's whole 512-entry register file), so nothing but the wave's own MFMAs can cover its
// softmax VALU work.  Round 4 measured "VALU between a wave's own 16-cycle MFMAs is paid in full" on a whole restructured kernel (profiles/
// r04_pipe1_nd1_negative.txt) — with packed FMAs, LDS scratch reads and DMA pieces in the same gaps.  This probe isolates the question: a loop of
// independent-accumulator MFMAs (the QK^T / PV pattern: a chain is revisited every 8th / 16th MFMA) with NF fillers of one kind behind every MFMA, one
// workgroup of four waves per CU on every CU (160 KiB of LDS requested: one workgroup per CU), random bf16 operands; s_memtime around the loop.
// Output: cycles per MFMA for every (kind, NF); the bare loop is the reference (16 cycles + issue).
//
//   hipcc --offload-arch=gfx950 -O2 tools/probes/gap_probe.hip -o tools/probes/bin/gap_probe && tools/probes/bin/gap_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;

enum Kind { K_NONE = 0, K_FMA, K_EXP, K_PKFMA, K_ADD, K_CVT, K_MAX, K_DSREAD, K_MIX_SOFTMAX, K_SALU, K_EXP_FMA, K_COUNT };
static const char* kNames[] = {"none", "v_fma_f32", "v_exp_f32", "v_pk_fma_f32", "v_add_f32", "v_cvt_pk_bf16_f32", "v_max_f32", "ds_read_b128", "softmax mix (fma,exp,add,cvt/2)",
                               "s_add_u32", "exp + fma alternating"};

template <int KIND>
__device__ __forceinline__ void filler(int i, float (&f)[16], f32x2 (&g)[4], uint32_t (&u)[4], f32x4 (&lr)[4], const __attribute__((address_space(3))) char* lp, uint32_t& sreg) {
  if constexpr (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i & 15]) : "v"(f[(i + 5) & 15]), "v"(f[(i + 9) & 15]));
  else if constexpr (KIND == K_EXP) asm volatile("v_exp_f32 %0, %1" : "=v"(f[i & 15]) : "v"(f[(i + 7) & 15]));
  else if constexpr (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(g[i & 3]) : "v"(g[(i + 1) & 3]), "v"(g[(i + 2) & 3]));
  else if constexpr (KIND == K_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i & 15]) : "v"(f[(i + 5) & 15]));
  else if constexpr (KIND == K_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i & 3]) : "v"(f[(i + 3) & 15]), "v"(f[(i + 11) & 15]));
  else if constexpr (KIND == K_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[i & 15]) : "v"(f[(i + 5) & 15]));
  else if constexpr (KIND == K_DSREAD) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(lr[i & 3]) : "v"(lp), "n"(0));
  else if constexpr (KIND == K_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sreg) : : "scc");
  else if constexpr (KIND == K_MIX_SOFTMAX) {
    // per score: half an FMA (exponent argument), one exp, one add (row sum), half a cvt — issued round-robin
    switch (i % 6) {
      case 0: asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i & 15]) : "v"(f[(i + 5) & 15]), "v"(f[(i + 9) & 15])); break;
      case 1: asm volatile("v_exp_f32 %0, %1" : "=v"(f[i & 15]) : "v"(f[(i + 7) & 15])); break;
      case 2: asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i & 15]) : "v"(f[(i + 5) & 15])); break;
      case 3: asm volatile("v_exp_f32 %0, %1" : "=v"(f[i & 15]) : "v"(f[(i + 7) & 15])); break;
      case 4: asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i & 15]) : "v"(f[(i + 5) & 15])); break;
      default: asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i & 3]) : "v"(f[(i + 3) & 15]), "v"(f[(i + 11) & 15])); break;
    }
  } else if constexpr (KIND == K_EXP_FMA) {
    if (i & 1) asm volatile("v_exp_f32 %0, %1" : "=v"(f[i & 15]) : "v"(f[(i + 7) & 15]));
    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i & 15]) : "v"(f[(i + 5) & 15]), "v"(f[(i + 9) & 15]));
  }
}

// NACC independent accumulator chains, visited round-robin (QK^T at D = 512: 8 chains; PV: 64)
template <int KIND, int NF, int NACC, int SHARE>
__global__ __launch_bounds__(256) void probe(const uint32_t* __restrict__ src, float* __restrict__ sink, unsigned long long* __restrict__ cycles, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const uint32_t* p = src + (size_t)(blockIdx.x * 256 + tid) * 64;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = *(const bf16x8*)(p + 4 * i);
    b[i] = *(const bf16x8*)(p + 16 + 4 * i);
  }
  float f[16];
  f32x2 g[4];
  uint32_t u[4] = {0, 0, 0, 0};
  f32x4 lr[4];
#pragma unroll
  for (int i = 0; i < 16; ++i) f[i] = __uint_as_float((p[32 + i] & 0x007fffffu) | 0x3f000000u);  // [0.5, 1)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    g[i] = f32x2{f[2 * i], f[2 * i + 1]};
    lr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int i = tid; i < 4096; i += 256) ((__attribute__((address_space(3))) uint32_t*)smem)[i] = p[i & 63];
  __syncthreads();
  const __attribute__((address_space(3))) char* lp = (const __attribute__((address_space(3))) char*)smem + (tid & 63) * 16;
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint32_t sreg = 0;
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 2 * NACC; ++m) {
      // SHARE = 2: one A fragment feeds two MFMAs (the kernels' two 16-row halves)
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m % NACC]) : "v"(a[(m / SHARE) & 3]), "v"(b[m & 3]));
#pragma unroll
      for (int k = 0; k < NF; ++k) filler<KIND>(m * NF + k, f, g, u, lr, lp, sreg);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_nop 15\n\ts_nop 3");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += f[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += g[i][0] + g[i][1] + __uint_as_float(u[i]) + lr[i][0] + lr[i][3];
  if (s == 1.2345f || sreg == 0xdeadbeefu) sink[blockIdx.x * 256 + tid] = s;
  if ((tid & 63) == 0) cycles[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int KIND, int NF, int NACC>
static void run(const uint32_t* src, float* sink, unsigned long long* cyc, int cus, double bare) {
  const int iters = 200;
  auto kern = probe<KIND, NF, NACC, 2>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(cus), dim3(256), 160 * 1024, 0, src, sink, cyc, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(cus * 4);
  hipMemcpy(h.data(), cyc, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
  double sum = 0;
  for (auto x : h) sum += (double)x;
  const double per = sum / h.size() / (iters * 2.0 * NACC);
  printf("GAP  chains %2d  %-34s x %d per MFMA: %6.2f cycles per MFMA", NACC, kNames[KIND], NF, per);
  if (bare > 0) printf("  (+ %5.2f over the bare loop = %5.2f per filler)", per - bare, NF ? (per - bare) / NF : 0.0);
  printf("\n");
}

template <int KIND, int NACC>
static void sweep(const uint32_t* src, float* sink, unsigned long long* cyc, int cus, double bare) {
  run<KIND, 1, NACC>(src, sink, cyc, cus, bare);
  run<KIND, 2, NACC>(src, sink, cyc, cus, bare);
  run<KIND, 3, NACC>(src, sink, cyc, cus, bare);
  run<KIND, 4, NACC>(src, sink, cyc, cus, bare);
}

static double bare_of(const uint32_t* src, float* sink, unsigned long long* cyc, int cus, int nacc) {
  const int iters = 200;
  auto launch = [&](auto kern, int NACC) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(cus), dim3(256), 160 * 1024, 0, src, sink, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(cus * 4);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto x : h) sum += (double)x;
    return sum / h.size() / (iters * 2.0 * NACC);
  };
  return nacc == 8 ? launch(probe<K_NONE, 0, 8, 2>, 8) : launch(probe<K_NONE, 0, 32, 2>, 32);
}

int main() {
  int cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount;
  const size_t n = (size_t)cus * 256 * 64;
  std::vector<uint32_t> h(n);
  srand(1);
  for (auto& x : h) {
    // two random bf16 in [-2, 2): sign, exponent 126 .. 128, random mantissa
    auto one = []() { return (uint32_t)(((rand() & 1) << 15) | ((126 + rand() % 3) << 7) | (rand() & 0x7f)); };
    x = one() | (one() << 16);
  }
  uint32_t* src;
  float* sink;
  unsigned long long* cyc;
  hipMalloc(&src, n * 4);
  hipMalloc(&sink, (size_t)cus * 256 * 4);
  hipMalloc(&cyc, (size_t)cus * 4 * 8);
  hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
  printf("# gap_probe on %s (%d CUs): one workgroup of four waves per CU, v_mfma_f32_16x16x32_bf16 on random operands, NF fillers behind every MFMA\n", prop.gcnArchName, cus);
  const double bare8 = bare_of(src, sink, cyc, cus, 8);
  printf("GAP  chains  8  bare loop: %6.2f cycles per MFMA\n", bare8);
  sweep<K_FMA, 8>(src, sink, cyc, cus, bare8);
  sweep<K_ADD, 8>(src, sink, cyc, cus, bare8);
  sweep<K_MAX, 8>(src, sink, cyc, cus, bare8);
  sweep<K_EXP, 8>(src, sink, cyc, cus, bare8);
  sweep<K_PKFMA, 8>(src, sink, cyc, cus, bare8);
  sweep<K_CVT, 8>(src, sink, cyc, cus, bare8);
  sweep<K_MIX_SOFTMAX, 8>(src, sink, cyc, cus, bare8);
  sweep<K_EXP_FMA, 8>(src, sink, cyc, cus, bare8);
  sweep<K_SALU, 8>(src, sink, cyc, cus, bare8);
  run<K_DSREAD, 1, 8>(src, sink, cyc, cus, bare8);
  const double bare32 = bare_of(src, sink, cyc, cus, 32);
  printf("GAP  chains 32  bare loop: %6.2f cycles per MFMA\n", bare32);
  sweep<K_FMA, 32>(src, sink, cyc, cus, bare32);
  sweep<K_EXP, 32>(src, sink, cyc, cus, bare32);
  sweep<K_MIX_SOFTMAX, 32>(src, sink, cyc, cus, bare32);
  return 0;
}
