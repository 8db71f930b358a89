// ffpa_philox.h — Philox4x32-10 at the logical score index and the dropout keep masks built from it (the reference's convention: prefill.cuh:398-546;
// the keep mask is pinned bit for bit by the GPU tests — against the CPU checker and against the reference's executed Triton dropout: tests/test_fwd_gpu.py).
#pragma once

#include "ffpa_common.h"

namespace ffpa {

// a ^ b ^ k, with X3 in ONE VALU instruction (gfx950: v_bitop3_b32, truth table 0x96 = three-input XOR; k is the wave-uniform round key in a scalar register).
// hipcc emits two v_xor_b32 for the expression: of the six VALU instructions of a Philox round two were this second XOR.  Round 5, the 16x16x32 dropout build
// without a bias: + 2.4 % at D = 512, + 3.0 % at D = 320, + 0.5 % at D = 1024, bit-identical (profiles/r05_philox_xor3.txt); the bias + dropout build keeps the plain
// form (with the asm its register allocation puts scratch accesses into the MFMA loops).
template <bool X3>
__device__ __forceinline__ uint32_t philox_xor3(uint32_t a, uint32_t b, uint32_t k) {
  if constexpr (X3) {
    uint32_t d;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(d) : "v"(a), "v"(b), "s"(k));
    return d;
  } else {
    return a ^ b ^ k;
  }
}

// Philox4x32-10 (Salmon et al., SC'11) on counter (quad_lo, quad_hi, 0, 0) with key = seed: the generator
// behind torch / cuRAND / Triton dropout.  The reference keys it by the logical score index so that masks
// line up with SDPA's (csrc/cuffpa/native/prefill.cuh:398-452): element e uses word e & 3 of block e >> 2.
template <bool X3 = false>
__device__ __forceinline__ void philox4x32_10(unsigned long long seed, unsigned long long quad, uint32_t (&out)[4]) {
  uint32_t c0 = (uint32_t)quad, c1 = (uint32_t)(quad >> 32), c2 = 0u, c3 = 0u;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a v_mul_hi_u32 + v_mul_lo_u32 pair: 32-bit integer
    // multiplies run at a quarter of the VALU rate and are what dropout costs
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    c0 = philox_xor3<X3>(hi1, c1, k0);
    c2 = philox_xor3<X3>(hi0, c3, k1);
    c1 = lo1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

// Element kept <=> u = ((float)word + 1.0f) * 2^-32 > p (prefill.cuh:437-440).  u is a monotone function of the 32-bit word, so the host
// resolves the float comparison ONCE per call into the smallest kept word (FwdArgs::keep_threshold, ffpa_capi.hip: dropout_keep_threshold) and
// the kernels compare integers: the same decision for every word, without a convert and a multiply-add per score.
// keep-scale (1/(1-p) or 0) for the 4 consecutive elements e0 .. e0+3 of one score row
__device__ __forceinline__ void dropout_keep4(unsigned long long seed, unsigned long long e0, uint32_t threshold, float keep_scale,
                                              float (&keep)[4]) {
  uint32_t blk[4];
  const unsigned a = (unsigned)(e0 & 3ull);
  philox4x32_10(seed, e0 >> 2, blk);
  if (__builtin_amdgcn_ballot_w64(a != 0) == 0ull) {
    // every lane's group is one whole Philox block (philox_offset and Nkv multiples of 4: the usual case): no second block,
    // no word selection
#pragma unroll
    for (int t = 0; t < 4; ++t) keep[t] = (blk[t] >= threshold) ? keep_scale : 0.f;
    return;
  }
  uint32_t w[8] = {blk[0], blk[1], blk[2], blk[3], 0, 0, 0, 0};
  if (a != 0) {  // the group straddles two Philox blocks
    philox4x32_10(seed, (e0 >> 2) + 1, blk);
#pragma unroll
    for (int i = 0; i < 4; ++i) w[4 + i] = blk[i];
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    // word a + t of the 8-word window, selected without run-time indexing (that would go to scratch)
    const uint32_t word = a == 0 ? w[t] : a == 1 ? w[t + 1] : a == 2 ? w[t + 2] : w[t + 3];
    keep[t] = (word >= threshold) ? keep_scale : 0.f;
  }
}

// The same decision as 4 bits (bit t <-> element e0 + t is kept): the 16x16x32 build draws the bits of a whole KV step before its
// exponentials and applies them when it packs P (the Philox temporaries are dead by then).  This form takes any element offset (a group that
// straddles two Philox blocks draws both): the rare case — philox_offset or Nkv not a multiple of 4.
template <bool X3 = false>
__device__ __forceinline__ uint32_t dropout_keep_bits4(unsigned long long seed, unsigned long long e0, uint32_t threshold) {
  uint32_t blk[4];
  const unsigned a = (unsigned)(e0 & 3ull);
  philox4x32_10<X3>(seed, e0 >> 2, blk);
  uint32_t w[8] = {blk[0], blk[1], blk[2], blk[3], 0, 0, 0, 0};
  if (a != 0) {  // the group straddles two Philox blocks
    philox4x32_10<X3>(seed, (e0 >> 2) + 1, blk);
#pragma unroll
    for (int i = 0; i < 4; ++i) w[4 + i] = blk[i];
  }
  uint32_t bits = 0u;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const uint32_t word = a == 0 ? w[t] : a == 1 ? w[t + 1] : a == 2 ? w[t + 2] : w[t + 3];
    bits |= (word >= threshold ? 1u : 0u) << t;
  }
  return bits;
}

// ... and the usual case, groups that are whole Philox blocks (element offset a multiple of 4), branch-free: N independent blocks advanced in
// lockstep, round by round (written out in the source because hipcc, at the register limit, keeps the source order).  Measured on the dropout
// workloads (profiles/r03_philox.txt): the branch-free form is + 3 ... 5 % over a wave-uniform branch per group; N = 2 / 4 add nothing — the
// cost is the VALU instruction count, not the latency of the multiply chain — so the kernels use N = 1.
template <int N, bool X3 = false>
__device__ __forceinline__ void dropout_keep_bits4_aligned_n(unsigned long long seed, const unsigned long long (&quad)[N], uint32_t threshold, uint32_t (&bits)[N]) {
  uint32_t c0[N], c1[N], c2[N], c3[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    c0[i] = (uint32_t)quad[i];
    c1[i] = (uint32_t)(quad[i] >> 32);
    c2[i] = 0u;
    c3[i] = 0u;
  }
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    uint32_t hi0[N], lo0[N], hi1[N], lo1[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0[i], p1 = (unsigned long long)0xCD9E8D57u * c2[i];
      hi0[i] = (uint32_t)(p0 >> 32);
      lo0[i] = (uint32_t)p0;
      hi1[i] = (uint32_t)(p1 >> 32);
      lo1[i] = (uint32_t)p1;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      c0[i] = philox_xor3<X3>(hi1[i], c1[i], k0);
      c2[i] = philox_xor3<X3>(hi0[i], c3[i], k1);
      c1[i] = lo1[i];
      c3[i] = lo0[i];
    }
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const uint32_t w[4] = {c0[i], c1[i], c2[i], c3[i]};
    uint32_t b = 0u;
#pragma unroll
    for (int t = 0; t < 4; ++t) b |= (w[t] >= threshold ? 1u : 0u) << t;
    bits[i] = b;
  }
}

}  // namespace ffpa
