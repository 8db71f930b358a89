// ffpa_fwd_m16_kernel.h — the prefill kernel mapped onto the 16x16x32 MFMA shape.
//
// Same algorithm, same pipeline (LDS-DMA K / V tiles, barriers A1 / A2 / B, lazy-rescale online softmax in the log2 domain)
// and same per-row recurrence as ffpa_fwd_split_d_kernel's prefill tiles (ffpa_fwd_kernel.h, which follows the
// reference's split_d_fwd_sm80: csrc/cuffpa/native/sm_80/split_d.cuh:96-777); only the mapping of the two products onto the
// matrix core differs.  Why a second mapping: v_mfma_f32_16x16x32 sustains a 14 - 21 % higher power-capped rate than
// v_mfma_f32_32x32x16 on random operands, and the D = 512 instruction mix built on it has a 9 % higher ceiling
// (tools/probes/stream_probe.hip, profiles/r02_stream_probe.txt, profiles/NOTES.md section 3).
//
// Mapping (one wave = 32 query rows as two 16-row halves rh; D <= 512: all of D per wave, 4 waves = 128 rows):
//   * S^T = K.Q^T per 16-key block kb and row half rh: A = K[16 keys][32 d] (one ds_read_b128 per lane: key lane % 16, d chunk
//     lane / 16), B = Q^T (resident fragments: row 16 rh + lane % 16, d chunk lane / 16).  One K fragment feeds BOTH row
//     halves, so LDS bytes per FLOP equal the 32x32x16 kernel's.  C layout: lane (n = lane % 16, c = lane / 16) holds keys
//     16 kb + 4 c + r (r < 4) of rows n and 16 + n.
//   * O^T += V^T.P^T per 16-column block db, 32-key step ks and row half: B = P^T straight from the S^T registers (contraction
//     slot 8 c + e <-> key 32 ks + 16 (e / 4) + 4 c + e % 4 — exactly what the lane holds), A = V^T by two
//     ds_read_b64_tr_b16 per lane (keys 32 ks + 4 c + .. and + 16), shared by both row halves.  O^T = 64 f32x4 tiles = the
//     256 AGPRs.
//   * a query row lives in 4 lanes (c = 0 .. 3): row max / row sum take two cross-lane steps (lane ^ 16, lane ^ 32).
//   * epilogue: a lane owns 4 consecutive columns of two rows; lanes c and c ^ 1 trade one group (v_permlane16_swap) so that
//     each stores whole 16-byte runs of ONE row.
// Tiles: D <= 512: 128 rows x 128 keys up to D = 320 (64 keys in the additive-bias build, whose LDS also holds the bias), 64 keys above;
// D > 512: D split over two waves, 64 rows x 32 keys, partial S^T tiles summed through LDS.  Every prefill launch at head dims
// >= FFPA_M16_MIN_D runs this kernel (ffpa_fwd_inst.hip): unmasked, boolean masks, additive biases, dropout, any softmax scale.
//
// Additive biases enter through the ACCUMULATOR: S^T(j) starts from bias / softmax_scale instead of zero, so that
// (K.Q^T + bias / scale) * scale = scale * K.Q^T + bias comes out of the same MFMA chain and the softmax below is the unmasked
// kernel's, instruction for instruction (prefill.cuh:556-658 adds the bias to the scaled score: same value up to fp32 rounding).
// Where the initial accumulators come from:
//   * a key bias (no row axis) is converted to fp32 / scale ONCE per workgroup into an LDS row cache: one ds_read_b128 per 16-key block;
//   * a bias with a row axis (fp16 / bf16 / fp32, unit key stride, 16-byte aligned rows) is staged through LDS by LDS-DMA one KV step
//     ahead: every wave fetches the [32 rows x BC keys] tile of step j + 1 into a private area between the PV MFMAs of step j (its own
//     queue drain at barrier B is all the synchronisation that needs) and converts it at the top of step j + 1;
//   * anything else (odd strides, boolean masks next to dropout) is read element by element from global memory (slow, rare).
#pragma once

#include "ffpa_fwd_kernel.h"

#ifndef FFPA_M16_MIN_D
// head dims from here up launch this kernel.  Round 2 (A/B against the 32x32x16 kernel, profiles/r02_m16_ab.txt): + 4 ... 5 % at D = 320 ... 512,
// + 0.5 ... 2.5 % at D = 576 ... 960, + 5 ... 6 % at D = 1024, even at D = 192 / 256, - 11 % at D = 128 -> 320.  Round 3, once the LDS-DMA
// destinations had become scalar base + immediate (no scalar register per piece: the small head dims had been spilling exactly those inside
// their loops): D = 256 1271 vs 1159 TFLOPS (+ 9.6 %), D = 192 1180 vs 1082 (+ 9 %), causal D = 256 + 8 %, N = 2048 + 6.6 %, dropout D = 256 + 22 %,
// key bias + 1.5 %, dense bias - 4 %; D = 128 - 14 % with 128-key tiles (64 score registers per lane) but, with 64-key tiles
// (FFPA_M16_BC128_MIN_D below), 1237 vs 1159 (+ 6.7 %), causal + 7.7 %, key bias + 46 %, dropout + 31 %, dense bias - 6 %, boolean mask at
// Nkv 2048 - 11 %; D = 64 - 5 % -> 128 (profiles/r03_m16_small_d.txt): the 32x32x16 prefill kernel is left with D = 64
#define FFPA_M16_MIN_D 128
#endif
#ifndef FFPA_M16_BC128_MIN_D
// 128-key tiles from this head dim up to FFPA_BC128_MAX_D (= 320: the LDS limit), 64-key tiles below: at D <= 192 the 128-key tile's 64 score
// registers per lane cost more than its fewer barriers return (D = 192: 1238 vs 1162 TFLOPS with 64 keys, D = 128: 1156 vs 923; D = 256: 1164 vs
// 1264, D = 320: 1250 vs 1307 — profiles/r03_m16_small_d.txt)
#define FFPA_M16_BC128_MIN_D 256
#endif
#ifndef FFPA_M16_PF1
#define FFPA_M16_PF1 6  // K fragments requested ahead of their (two) MFMAs
#endif
#ifndef FFPA_M16_PF2
#define FFPA_M16_PF2 4  // V^T fragments (two transpose reads each) requested ahead
#endif
#ifndef FFPA_M16_K_PRE
#define FFPA_M16_K_PRE 8  // K(j+1) pieces issued between the softmax stages (a multiple of 4); the rest go out between the PV MFMAs
#endif
#ifndef FFPA_M16_PF_WAVES
#define FFPA_M16_PF_WAVES 2  // L2 prefetch: waves of a workgroup that touch (2: one K and one V slice per step; 4: two of each)
#endif
#ifndef FFPA_M16_PF_WHICH
#define FFPA_M16_PF_WHICH 3  // L2 prefetch: bit 0 = K, bit 1 = V
#endif
#ifndef FFPA_M16_PF_DIST
#define FFPA_M16_PF_DIST 2  // L2 prefetch: steps ahead of the step being computed (the DMA queue itself covers 1)
#endif
#ifndef FFPA_M16_K_PRE_ND2
#define FFPA_M16_K_PRE_ND2 64  // ditto for the split-D tiles (D > 512; clamped to the tile's pieces: all of K(j+1) goes out between the softmax stages)
#endif
// ---- split-D tiles (D > 512), round 4: the K tile in two 16-key halves (K1 = keys 0 .. 15, K2 = keys 16 .. 31 of a tile: each wave stages PPW / 2 pieces
// of either), so that the rows of one half can take the next tile's pieces while the other half is still being contracted — what the softmax pipeline
// below is built on.  (The intermediate schedule of round 4 — two-half K with a fourth barrier, no pipeline: + 3.9 % where the pipeline gives + 6.5 % —
// lives on as tools/experiments/r05_pruned_switches.diff.)
// ---- the SOFTMAX PIPELINE of the split-D tiles.  With the K tile in two 16-key halves, key block 0 of tile
// j + 1 can be contracted while the softmax of tile j runs: the step becomes
//     Q: QK^T(j) key block 1                         | DMA: K1(j+2) rest, V(j) first part
//     barrier A1 (partial S^T of block 1 visible, K2(j) rows free, K1(j+1) landed)
//     S: softmax(j), one small group of VALU instructions in every gap of the 32 MFMAs of QK^T(j+1) key block 0   | DMA: V(j) rest, K2(j+1) first part
//     barrier A2 (V(j) landed, K1(j+1) rows free)
//     P: PV(j)                                       | DMA: K2(j+1) rest, K1(j+2) first part
//     barrier B (V(j) rows free, K2(j+1) landed)
// — three barriers again, the latency chain of the softmax (partial-S exchange, row max over four lanes, exponentials: ~ 800 cycles during
// which the matrix core idled) covered by 512 cycles of MFMA work, and every DMA stream has two phases to be issued in.  The partial S^T of
// block 0 is published one step early, so its exchange area is double-buffered (6 KiB per wave instead of 4).  Piece counts per phase in
// sixteenths of a wave's pieces per tile image: FFPA_M16_PP_VQ (V pieces in Q; the rest in S), _K1Q (K1 pieces in Q; the rest in the P phase
// before), _K2S (K2 pieces in S; the rest in P).  Builds: no additive bias, no dropout (MK 0 and, since round 5, the boolean-mask / mask-range build MK 2); D % 128 == 0.
// Measured (config 3, B1 H32 N8192 D1024, interleaved A/B on one box, outputs bit-identical; profiles/r04_pipe.txt): round-3 loop 945 ... 975 TFLOPS,
// two-half K alone (best table) + 3.9 %, the pipeline + 6.2 ... 6.6 % (VQ / K1Q / K2S = 6 / 2 / 2, three K fragments ahead in S;
// 7 / 1 / 1: + 5 %, 8 / 0 / 0: + 4.2 %, 10 / 1 / 1: + 1.8 %); D = 640 / 768 / 896: + 4.4 / + 7.6 / + 3.4 %, causal + 4.7 %, Nq 1024 + 9.9 %, 32k keys + 3.5 %.
#ifndef FFPA_M16_PP_VQ
#define FFPA_M16_PP_VQ 6
#endif
#ifndef FFPA_M16_PP_K1Q
#define FFPA_M16_PP_K1Q 2
#endif
#ifndef FFPA_M16_PP_K2S
#define FFPA_M16_PP_K2S 2
#endif
#ifndef FFPA_M16_PP_QSTEP
#define FFPA_M16_PP_QSTEP 2  // one piece every this many fragments from the phase's start (0: spread evenly over the phase); Q: 2 vs 1 + 1.6 % on config 3
#endif
#ifndef FFPA_M16_PP_SSTEP
#define FFPA_M16_PP_SSTEP 1
#endif
#ifndef FFPA_M16_PP_PSTEP
#define FFPA_M16_PP_PSTEP 2
#endif
// Where the softmax's instruction groups sit in the S phase (measured, profiles/r04_pipe.txt: at D = 1024 "one per MFMA gap" and "all behind the phase's MFMAs" run equally
// fast — VALU between a wave's own MFMAs is paid in full either way —, at D = 640 behind is + 1.9 %, in front - 3 ... 4 % everywhere): behind for D < 1024 and for the mask build
// (whose softmax carries a branch — the mask read; per gap it cost D = 1024 3 ... 5 %: profiles/r05_mask_pipeline.txt), one group per gap at D = 1024.
#ifndef FFPA_M16_PP_PF
#define FFPA_M16_PP_PF 3  // K fragments requested ahead of their MFMAs in the S phase (the softmax's registers are live next to them)
#endif
// A DMA piece and the MFMA in front of it are one asm statement (Mfma16::with_dma): the MFMA is the wait state between the M0 write and the piece,
// no s_nop.  Interleaved A/B against separate statements, bit-identical (profiles/r04_pipe.txt): config 2 + 1.0 %, cross + 1.7 %, causal + 1.3 %, D = 320 + 1.4 %, config 4 + 0.9 %, D = 1024 +- 0


// Developer instrumentation (-DFFPA_M16_TIMING, tools/gpu_phase_times.py; never in the shipped build): every wave accumulates the shader
// clock cycles it spends in six phases of the KV-tile loop and lane 0 writes the totals over the LSE of its first rows.
#ifdef FFPA_M16_TIMING
#define FFPA_TSTAMP(i)                                         \
  do {                                                         \
    __builtin_amdgcn_sched_barrier(0);                         \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    tacc[i] += t_ - tprev;                                     \
    tprev = t_;                                                \
    __builtin_amdgcn_sched_barrier(0);                         \
  } while (0)
#else
#define FFPA_TSTAMP(i) \
  do {                 \
  } while (0)
#endif

namespace ffpa {

// Keys per KV tile of ffpa_fwd_m16_kernel<., D, MK>: 32 for the split-D head dims, 64 for the additive-bias builds (their LDS also holds the
// bias) and for head dims outside [FFPA_M16_BC128_MIN_D, FFPA_BC128_MAX_D], 128 inside.  The launch plan (ffpa_capi.hip) uses the same rule.
constexpr int m16_block_keys(int D, bool bias_build) {
  return D > 512 ? 32 : ((!bias_build && D >= FFPA_M16_BC128_MIN_D && D <= FFPA_BC128_MAX_D) ? 128 : 64);
}

// Bytes of the partial-S^T exchange area behind the two tile images of the split-D tiles (D > 512): 4 KiB per wave; 6 KiB in the builds without
// an additive bias, whose softmax pipeline publishes key block 0 one step early into a double-buffered half (reserved whether or
// not the build is pipelined, so that the launch side needs to know the mask kind only).
// Since round 5 the two waves of a row block share the softmax by rows and trade P^T fragments (4 x 1 KiB) and per-row scalars (4 x 256 B) behind it: 5 KiB more
// in the builds without an additive bias.  MK: 1 = the additive-bias build with staged tiles (4 KiB per wave), 3 = the key-bias build: 6 KiB where it runs the
// softmax pipeline (D % 128 == 0; its bias lives in a ring cache behind the exchange, and its P^T / scalar slots are slots of a wave's own partials that only
// the wave itself reads back: D = 1024 has no 5 KiB to spare next to the ring — the own-slot form measured 0.5 ... 1 % slower on the unmasked build), 4 KiB elsewhere.
constexpr int m16_exchange_bytes(int D, int MK) {
  return D > 512 ? ((MK == 1 || (MK == 3 && D % 128 != 0)) ? 4 * 4096 : (MK == 3 ? 4 * 6144 : 4 * 6144 + 4 * 1024 + 4 * 256)) : 0;
}

// Which of `cnt` DMA pieces, if any, rides on fragment n of a loop of N fragments — piece t sits on
// fragment t * step (step > 0: front-loaded) or floor(t N / cnt) (step == 0: spread evenly); -1 = none.
constexpr int m16_piece_at(int n, int N, int cnt, int step) {
  for (int t = 0; t < cnt; ++t)
    if ((step > 0 ? t * step : (t * N) / cnt) == n) return t;
  return -1;
}

// The MFMAs are inline asm: the S^T accumulators must be VGPRs and the O^T tiles exactly the 256 AGPRs, in place (left to hipcc,
// parts of O^T end up in VGPRs and the Q fragments in scratch); first / acc: S^T (VGPR form), acc_a: O^T (AGPR form).  The
// operands come from ds_read / global loads / v_cvt long before: tools/check_mfma_hazards.py checks the generated ISA.
template <typename T>
struct Mfma16;
template <>
struct Mfma16<__bf16> {
  typedef Elem<__bf16>::v8 v8;
  static __device__ __forceinline__ void first(f32x4& d, v8 a, v8 b) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b)); }
  static __device__ __forceinline__ void acc(f32x4& d, v8 a, v8 b) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b)); }
  static __device__ __forceinline__ void acc_a(f32x4& d, v8 a, v8 b) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b)); }
  // the LAST MFMA of a pair of accumulator chains, with the MFMA-result -> VALU-reader wait states in the same statement: `other` (the chain
  // that ended one MFMA earlier) is an operand too, so nothing the compiler emits — a register copy for a loop-carried value included — can read
  // either accumulator before the pad (a separate pad statement does not bind the copies hipcc places in front of it)
  static __device__ __forceinline__ void acc_last(f32x4& d, f32x4& other, v8 a, v8 b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\ts_nop 15\n\ts_nop 3" : "+v"(d), "+v"(other) : "v"(a), "v"(b));
  }
  // An MFMA with a 1 KiB LDS-DMA piece riding on it: M0 is written in FRONT of the MFMA, which then is the wait state an
  // LDS-DMA needs behind an M0 write — the piece costs the stream two issue slots instead of three (s_add, s_nop, buffer_load).
  // KIND 0: first MFMA of a chain (C = 0, VGPR), 1: accumulate (VGPR), 2: accumulate (AGPR tile).
  template <int KIND, int LCONST>
  static __device__ __forceinline__ void with_dma(f32x4& d, v8 a, v8 b, u32x4 rsrc, uint32_t lds_base, uint32_t voff, uint32_t soff) {
    if constexpr (KIND == 0)
      asm volatile("s_add_u32 m0, %3, %7\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0\n\tbuffer_load_dwordx4 %4, %5, %6 offen lds"
                   : "=&v"(d) : "v"(a), "v"(b), "s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff), "n"(LCONST) : "memory", "scc" FFPA_M0_CLOBBER);
    else if constexpr (KIND == 1)
      asm volatile("s_add_u32 m0, %3, %7\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tbuffer_load_dwordx4 %4, %5, %6 offen lds"
                   : "+v"(d) : "v"(a), "v"(b), "s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff), "n"(LCONST) : "memory", "scc" FFPA_M0_CLOBBER);
    else
      asm volatile("s_add_u32 m0, %3, %7\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tbuffer_load_dwordx4 %4, %5, %6 offen lds"
                   : "+a"(d) : "v"(a), "v"(b), "s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff), "n"(LCONST) : "memory", "scc" FFPA_M0_CLOBBER);
  }
};
template <>
struct Mfma16<_Float16> {
  typedef Elem<_Float16>::v8 v8;
  static __device__ __forceinline__ void first(f32x4& d, v8 a, v8 b) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b)); }
  static __device__ __forceinline__ void acc(f32x4& d, v8 a, v8 b) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b)); }
  static __device__ __forceinline__ void acc_a(f32x4& d, v8 a, v8 b) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b)); }
  static __device__ __forceinline__ void acc_last(f32x4& d, f32x4& other, v8 a, v8 b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\ts_nop 15\n\ts_nop 3" : "+v"(d), "+v"(other) : "v"(a), "v"(b));
  }
  // An MFMA with a 1 KiB LDS-DMA piece riding on it: M0 is written in FRONT of the MFMA, which then is the wait state an
  // LDS-DMA needs behind an M0 write — the piece costs the stream two issue slots instead of three (s_add, s_nop, buffer_load).
  // KIND 0: first MFMA of a chain (C = 0, VGPR), 1: accumulate (VGPR), 2: accumulate (AGPR tile).
  template <int KIND, int LCONST>
  static __device__ __forceinline__ void with_dma(f32x4& d, v8 a, v8 b, u32x4 rsrc, uint32_t lds_base, uint32_t voff, uint32_t soff) {
    if constexpr (KIND == 0)
      asm volatile("s_add_u32 m0, %3, %7\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, 0\n\tbuffer_load_dwordx4 %4, %5, %6 offen lds"
                   : "=&v"(d) : "v"(a), "v"(b), "s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff), "n"(LCONST) : "memory", "scc" FFPA_M0_CLOBBER);
    else if constexpr (KIND == 1)
      asm volatile("s_add_u32 m0, %3, %7\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\tbuffer_load_dwordx4 %4, %5, %6 offen lds"
                   : "+v"(d) : "v"(a), "v"(b), "s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff), "n"(LCONST) : "memory", "scc" FFPA_M0_CLOBBER);
    else
      asm volatile("s_add_u32 m0, %3, %7\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\tbuffer_load_dwordx4 %4, %5, %6 offen lds"
                   : "+a"(d) : "v"(a), "v"(b), "s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff), "n"(LCONST) : "memory", "scc" FFPA_M0_CLOBBER);
  }
};

// A query row lives in the 4 lanes n, n + 16, n + 32, n + 48, and every lane carries a value for two rows (n and 16 + n).
// Both 4-lane reductions together in three register swaps (no LDS crossbar): v_permlane32_swap pairs row n's halves in lanes
// 0 .. 31 and row 16 + n's in lanes 32 .. 63, v_permlane16_swap folds the remaining lane ^ 16 step, and a last
// v_permlane32_swap hands every lane both results.
template <bool IS_MAX>
__device__ __forceinline__ void row4_reduce2(float& t0, float& t1) {
  auto op = [](float x, float y) { return IS_MAX ? fmaxf(x, y) : x + y; };
  const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t0), __float_as_uint(t1), false, false);
  const float v = op(__uint_as_float(s1[0]), __uint_as_float(s1[1]));
  const auto s2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float w = op(__uint_as_float(s2[0]), __uint_as_float(s2[1]));
  const auto s3 = __builtin_amdgcn_permlane32_swap(__float_as_uint(w), __float_as_uint(w), false, false);
  t0 = __uint_as_float(s3[0]);
  t1 = __uint_as_float(s3[1]);
}

// one value per lane, reduced over the 4 lanes (n, n + 16, n + 32, n + 48) of a query row: two register swaps, in row4_reduce2's order
// (lane ^ 32 first, then lane ^ 16: a row sum comes out in the same bits whichever of the two reduced it)
template <bool IS_MAX>
__device__ __forceinline__ void row4_reduce1(float& t) {
  auto op = [](float x, float y) { return IS_MAX ? fmaxf(x, y) : x + y; };
  const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
  const float v = op(__uint_as_float(s1[0]), __uint_as_float(s1[1]));
  const auto s2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  t = op(__uint_as_float(s2[0]), __uint_as_float(s2[1]));
}
// LDS images of the K / V tiles: row-major [BC][D], 16-byte slot s of row `key` stored at slot s ^ swizzle(key) (applied on the
// DMA's per-lane source offset).  K fragments are fetched by ds_read_b128 whose 16-lane groups hold 16 different keys and two
// neighbouring slots; V^T fragments by ds_read_b64_tr_b16 whose 32-lane halves hold 8 keys x 32 bytes.  Row strides that are whole
// 256-byte bank rows (D % 128 == 0) need the full 4-bit / 3-bit spread, the others (D % 128 == 64: consecutive rows already sit
// half a bank row apart) one bit less — and their rows only have room for a 3-bit XOR (D / 8 slots, a multiple of 8).
// tools/sim_lds_layout.py replays both maps and the bank rule on the host (tests/test_lds_layout.py).
template <int D>
__device__ __forceinline__ int m16_k_swizzle(int key) {
  return (D % 128 == 0) ? (key & 15) : ((key >> 1) & 7);
}
template <int D>
__device__ __forceinline__ int m16_v_swizzle(int key) {
  return (D % 128 == 0) ? ((key & 7) << 1) : (((key >> 1) & 3) << 1);
}

// The product library is built with -DFFPA_PRODUCT_BUILD (ffpa_attn_amd/build.py: libffpa_attn_hip.so and its test twin): none of the
// developer switches of this file and of ffpa_fwd_kernel.h — ablations that compute WRONG results, cycle-counter instrumentation that
// overwrites LSE rows, experimental schedules — may be anything but its shipped default there.  Variant libraries (build.py --variant)
// are built without the define, get another file name and say so in ffpa_attn_version().
#ifdef FFPA_PRODUCT_BUILD
#if defined(FFPA_M16_TIMING) || FFPA_M16_PF1 != 6 || FFPA_M16_PF2 != 4 || FFPA_M16_K_PRE != 8 || \
    FFPA_M16_K_PRE_ND2 != 64 || FFPA_M16_PF_DIST != 2 || FFPA_M16_PF_WAVES != 2 || FFPA_M16_PF_WHICH != 3 || FFPA_M16_MIN_D != 128 || FFPA_M16_BC128_MIN_D != 256 || \
    FFPA_M16_PP_VQ != 6 || FFPA_M16_PP_K1Q != 2 || FFPA_M16_PP_K2S != 2 || FFPA_M16_PP_QSTEP != 2 || \
    FFPA_M16_PP_SSTEP != 1 || FFPA_M16_PP_PSTEP != 2 || FFPA_M16_PP_PF != 3
#error "FFPA_PRODUCT_BUILD: a developer switch is not at its shipped default"
#endif
#endif

// MK (mask kind): 0 = the build for calls without attn_bias / mask ranges, 2 = boolean masks (FFPA_BIAS_BOOL8 bytes and / or kv_bounds
// ranges: what ffpa_attn_func(attn_mask=<bool>) launches), 1 = additive biases (fp16 / bf16 / fp32, any broadcast; boolean masks too when
// they come with dropout), 3 = key biases only (no row axis: [B|1, H|1, 1, Nkv], the reference bench's "attn-mask" case, key padding as an
// additive mask) from the LDS row cache — the bias-free kernel plus one LDS read per 16-key block and row half, nothing else rides along.
// DROP: the dropout-capable builds (Philox4x32-10 at the logical score index, applied to the rounded P:
// prefill.cuh:398-546) — of the bias-free kernel (MK = 0: nothing but the Philox code rides along) and of the additive-bias kernel.
template <typename T, int D, int MK = 0, bool DROP = false>
__global__ __launch_bounds__(256) void ffpa_fwd_m16_kernel(const FwdArgs a) {
  static_assert(MK >= 0 && MK <= 3 && (!DROP || MK <= 1), "mask kinds 0 ... 3; dropout builds: MK = 0 (no bias) and MK = 1 (any bias or mask)");
  constexpr bool MASK = MK == 1 || MK == 2;  // builds that honour mask ranges (kv_bounds)
  constexpr bool kBias = MK == 1 || MK == 3;  // builds whose S^T accumulators start from bias / scale
  using E = Elem<T>;
  using M = Mfma16<T>;
  using v8 = typename E::v8;
  using v4 = typename E::v4;
  static_assert(D % 64 == 0 && D <= 1024, "O^T (D / ND / 2 registers per lane) must fit the AGPRs");
  // D > 512: the head dim is split over two waves (ND = 2, as in ffpa_fwd_split_d_kernel): wave (qb, dh) = (wave / 2, wave % 2) owns rows
  // 32 qb .. + 32 and columns dh * D/2 .. of both contractions; the two partial S^T tiles of a row block are summed through LDS.
  constexpr int ND = (D <= 512) ? 1 : 2;
  constexpr int DW = D / ND;    // columns owned by one wave
  constexpr int BC = m16_block_keys(D, kBias), BR = 128 / ND;
  constexpr int KS = DW / 32;   // QK contraction steps per wave
  constexpr int NKB = BC / 16;  // 16-key S^T blocks per tile
  constexpr int NKS = BC / 32;  // PV contraction steps per tile
  constexpr int NDB = DW / 16;  // 16-column O^T blocks per wave
  constexpr int RB = D * 2;
  constexpr int TILE = BC * RB;
  constexpr int PPW = BC * D * 2 / 4096;  // 1 KiB DMA pieces per wave per tile
  // D = 512 with masks: a tile row is one whole piece -> wave-uniform rows, scalar addressing (the per-lane offset tables of the other form
  // would not fit next to the mask path's registers).  Everything else keeps tile-invariant per-lane offsets in registers: measured equal
  // or better (D = 512 unmasked: + 0 ... 2 %; D = 1024: 927 vs 776 TFLOPS — the scalar row form loses 16 % there on this build).
  constexpr bool kRowDma = RB % 1024 == 0 && ND == 1 && (MK == 1 || MK == 2 || DROP);
  static_assert(!kRowDma || RB == 1024, "the scalar row form is used where a tile row is exactly one piece");
  constexpr int KPW = BC / 4;                // keys staged per wave per tile
  constexpr int PF1 = FFPA_M16_PF1, PF2 = FFPA_M16_PF2;
  constexpr int kPreReq = ND == 2 ? FFPA_M16_K_PRE_ND2 : FFPA_M16_K_PRE;
  constexpr int kPre = ((kPreReq < PPW ? kPreReq : PPW) / 4) * 4;
  constexpr int N1 = KS * NKB;   // K fragments per tile
  constexpr int N2 = NDB * NKS;  // V^T fragments per tile
  // the softmax pipeline of the split-D tiles (header comment above the FFPA_M16_PP_* piece counts): needs an even number of pieces per wave (D % 128 == 0), and it stages K as two
  // 16-key halves — piece i < kH of a wave belongs to K1 (keys 0 .. 15), the rest to K2
  constexpr bool kPipe = ND == 2 && PPW % 2 == 0 && NKB == 2 && (MK == 0 || MK == 2 || MK == 3) && !DROP;
  constexpr bool kKS = kPipe;  // the two-half K piece map
  constexpr int kH = PPW / 2;  // K1 / K2 pieces per wave
  constexpr int ppVQ = PPW * FFPA_M16_PP_VQ / 16, ppVS = PPW - ppVQ;     // V(j): in Q(j), in S(j)
  constexpr int ppK1Q = PPW * FFPA_M16_PP_K1Q / 16, ppK1P = kH - ppK1Q;  // K1(j+1): in P(j-1) (first), in Q(j) (rest)
  constexpr int ppK2S = PPW * FFPA_M16_PP_K2S / 16, ppK2P = kH - ppK2S;  // K2(j+1): in S(j) (first), in P(j) (rest)
  static_assert(!kPipe || (ppVQ >= 0 && ppVS >= 0 && ppK1Q >= 0 && ppK1P >= 0 && ppK2S >= 0 && ppK2P >= 0), "piece counts per phase");
  static_assert(!kPipe || (ppVQ + ppK1Q <= KS && ppVS + ppK2S <= KS && ppK2P + ppK1P <= N2), "at most one piece per fragment");
  // (a step that walks past the phase's last fragment would silently drop the pieces behind it: a tile image with stale rows)
  static_assert(!kPipe || ((ppVQ + ppK1Q - 1) * FFPA_M16_PP_QSTEP < KS && (ppVS + ppK2S - 1) * FFPA_M16_PP_SSTEP < KS && (ppK2P + ppK1P - 1) * FFPA_M16_PP_PSTEP < N2),
                "every piece of a phase must ride on one of its fragments");
  // counted waits: order of issue inside a step — Q: K1 rest, V first; S: V rest, K2 first; P: K2 rest, K1 first (of the tile after next), touch
  constexpr int ppWaitA1 = ppVQ + (ppK1Q > 0 ? 0 : 0);  // K1(j+1) has landed: the V pieces of Q(j) stay in flight (+ the touch when K1Q == 0)
  constexpr int ppWaitA2 = ppK2S;                       // V(j) has landed
  constexpr int ppWaitB = ppK1P;                        // K2(j+1) has landed (+ the touch)
  constexpr int kStep1 = N1 / PPW > 0 ? N1 / PPW : 1;  // one V piece every this many K fragments
  constexpr int kStep2 = N2 / PPW > 0 ? N2 / PPW : 1;  // one K piece every this many V^T fragments
  static_assert(kStep1 >= 1 && kStep2 >= 1 && N1 % PPW == 0 && N2 % PPW == 0, "DMA pieces must fit the MFMA loops");
  constexpr int NH = BC > 64 ? BC / 64 : 1;         // 64-key halves of a tile (ds_read immediates are 16 bits: one address base per half)
  constexpr int KV = (D % 128 == 0) ? 4 : 2;        // K fragment address variants: the swizzle reaches slot bits 0 .. 3 / 0 .. 2
  constexpr int KVB = (D % 128 == 0) ? 256 : 128;   //   and the bytes KV contraction steps advance
  constexpr int VV = (D % 128 == 0) ? 8 : 4;        // V^T fragment address variants
  constexpr int VVB = (D % 128 == 0) ? 256 : 128;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  FFPA_LDS char* const Kt = (FFPA_LDS char*)smem;
  FFPA_LDS char* const Vt = Kt + TILE;
  FFPA_LDS char* const Xb = Kt + 2 * TILE;  // ND == 2: partial-S exchange, 4 KiB per wave (6 KiB in the builds without an additive bias)
  FFPA_LDS char* const Bl = Kt + 2 * TILE + m16_exchange_bytes(D, MK);  // key-bias row cache (FwdArgs.bias_lds bytes, when enabled)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15;
  const int c = lane >> 4;
  const int qb = wave / ND;  // row block of this wave
  const int dh = wave % ND;  // which D / ND slice of the head dim it owns

  // workgroup -> (batch, head, row tile, split): as ffpa_fwd_split_d_kernel (all row tiles of a head on one XCD)
  int vid = blockIdx.x;
  if (!(a.flags & kFlagNoXcdRemap)) vid = xcd_logical_id(vid, gridDim.x, a.xcd_group);
  const int split = vid % a.nsplit;
  vid /= a.nsplit;
  // PAIRED ROW TILES (FwdArgs::pair_tiles, round 6: launches under the causal flag): workgroup i of a head walks row tile nqt - 1 - i (its long one) and then
  // row tile i (the short one) — every workgroup of the launch walks nqt + 1 diagonal-bounded tiles' worth of KV steps, the launch has half the workgroups
  // (half the dispatches, no tail of short workgroups), and per row nothing changes: the same tile, the same recurrence, the same bits.  The whole body
  // below — tile range, Q fragments, KV loop, epilogue — runs once per pass; LDS is free between the passes (every wave has passed the last step's barrier
  // B, and the pipelined loop's own post-loop barrier, before any wave can start the next pass's prologue DMA).
  const int nqt_wg = a.pair_tiles ? (a.nqt + 1) >> 1 : a.nqt;
  const int bh = vid / nqt_wg;
  const int qt_wg = vid - bh * nqt_wg;
  const int npass = (a.pair_tiles && 2 * qt_wg != a.nqt - 1) ? 2 : 1;  // (an odd tile count: the middle tile is its own partner)
  for (int pass = 0; pass < npass; ++pass) {
  int qt = qt_wg;
  // longest rows first: with the causal flag — and with mask ranges, whose usual source is a causal-like boolean mask (later rows see
  // more keys; for any other mask the order of a head's row tiles does not matter) — so that the launch ends on its short workgroups
  if (a.pair_tiles) qt = pass == 0 ? a.nqt - 1 - qt_wg : qt_wg;
  else if (a.causal || (MASK && a.kv_bounds != nullptr)) qt = a.nqt - 1 - qt;
  const int b = bh / a.Hq;
  const int hq = bh - b * a.Hq;
  const int hkv = hq / a.group;
  const int q0 = qt * BR;
  const int wq0 = q0 + qb * 32;
  int qrow[2], qrow_c[2];
#pragma unroll
  for (int rh = 0; rh < 2; ++rh) {
    qrow[rh] = wq0 + 16 * rh + n16;
    qrow_c[rh] = qrow[rh] < a.Nq ? qrow[rh] : a.Nq - 1;
  }

  const T* __restrict__ Kg = (const T*)a.k + b * a.sk[0] + hkv * a.sk[1];
  const T* __restrict__ Vg = (const T*)a.v + b * a.sv[0] + hkv * a.sv[1];
  const uint32_t k_row_bytes = (uint32_t)a.sk[2] * 2u;
  const uint32_t v_row_bytes = (uint32_t)a.sv[2] * 2u;

  // ---- LDS-DMA.  A caller's head dim below D (a multiple of 8): K columns at and past it read as zeros (lanes whose source slot
  // lies there get an out-of-range offset: the descriptor's range check zero-fills them), Q columns are not loaded, O columns are
  // not stored.  Two addressing forms:
  //   * D = 512 builds with a mask path (a row = one piece): wave w stages keys 16 a + 4 w + b4, lane l -> slot l of the row;
  //     everything but the swizzled lane offset is scalar;
  //   * other head dims: piece p = wave * PPW + i covers slots [64 p, 64 p + 64) of the row-major image; the per-lane source
  //     offsets are tile-invariant and live in PPW + PPW registers.
  const uint32_t rb_valid = (uint32_t)a.d_valid * 2u;
  const int slots_valid = a.d_valid >> 3;
  uint32_t kvo[kRowDma ? 4 : 1], vvo[kRowDma ? 4 : 1];
  uint32_t kro[kRowDma ? KPW : 1], vro[kRowDma ? KPW : 1];
  uint32_t krel[kRowDma ? 1 : PPW], vrel[kRowDma ? 1 : PPW];
  uint32_t k_lds = 0, v_lds = 0;
  if constexpr (kRowDma) {
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
      kvo[bb] = (uint32_t)((lane ^ m16_k_swizzle<D>(4 * wave + bb)) << 4);
      if ((lane ^ m16_k_swizzle<D>(4 * wave + bb)) >= slots_valid) kvo[bb] = kDmaOob;
      vvo[bb] = (uint32_t)((lane ^ m16_v_swizzle<D>(4 * wave + bb)) << 4);
      if ((lane ^ m16_v_swizzle<D>(4 * wave + bb)) >= slots_valid) vvo[bb] = kDmaOob;  // (V too: O^T columns past the head dim stay exact zeros)
    }
#pragma unroll
    for (int jk = 0; jk < KPW; ++jk) {
      const uint32_t key = (uint32_t)(16 * (jk >> 2) + 4 * wave + (jk & 3));
      kro[jk] = key * k_row_bytes;
      vro[jk] = key * v_row_bytes;
    }
    k_lds = (uint32_t)(uintptr_t)Kt + (uint32_t)(4 * wave * RB);
    v_lds = (uint32_t)(uintptr_t)Vt + (uint32_t)(4 * wave * RB);
  } else {
    constexpr int SPR = D / 8;  // 16-byte slots per row
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int g = (wave * PPW + i) * 64 + lane;
      const int key = g / SPR;
      const int slot = g - key * SPR;
      const int vs = slot ^ m16_v_swizzle<D>(key);
      vrel[i] = (uint32_t)key * v_row_bytes + (uint32_t)(vs << 4);
      if (vs >= slots_valid) vrel[i] = kDmaOob;  // (V too: O^T columns past the head dim stay exact zeros)
      // K: the same pieces, or (two-half schedule) this wave's pieces i < PPW / 2 from the first 16 keys and the rest from the last 16:
      // piece i of wave w is piece w PPW/2 + i of K1 (the image's first 2 PPW KiB), resp. w PPW/2 + i - PPW/2 of K2
      const int gk = kKS ? ((i < kH ? wave * kH + i : 2 * PPW + wave * kH + (i - kH)) * 64 + lane) : g;
      const int kkey = gk / SPR;
      const int kslot = gk - kkey * SPR;
      const int ks = kslot ^ m16_k_swizzle<D>(kkey);
      krel[i] = (uint32_t)kkey * k_row_bytes + (uint32_t)(ks << 4);
      if (ks >= slots_valid) krel[i] = kDmaOob;
    }
    // this wave's pieces land at base + i KiB: one scalar base per tile image, the piece index is an immediate of the DMA asm
    k_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(uintptr_t)Kt + (uint32_t)(wave * (kKS ? kH : PPW) * 1024)));
    v_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(uintptr_t)Vt + (uint32_t)(wave * PPW * 1024)));
  }
  auto issue_k = [&](auto ic, int key0) {
    constexpr int i = decltype(ic)::value;
    const TileSrc ts = tile_src<BC>(Kg, k_row_bytes, key0, a.Nkv, rb_valid);
    if constexpr (kRowDma) {
      lds_dma_row<(16 * (i >> 2) + (i & 3)) * RB, 0>(ts.rsrc, k_lds, kvo[i & 3], kro[i]);
    } else {
      lds_dma_16_at<(kKS && i >= kH ? 2 * PPW + (i - kH) : i) * 1024>(ts.rsrc, k_lds, krel[i], 0u);
    }
  };
  auto issue_v = [&](auto ic, int key0) {
    constexpr int i = decltype(ic)::value;
    const TileSrc ts = tile_src<BC>(Vg, v_row_bytes, key0, a.Nkv, rb_valid);
    if constexpr (kRowDma) {
      lds_dma_row<(16 * (i >> 2) + (i & 3)) * RB, 0>(ts.rsrc, v_lds, vvo[i & 3], vro[i]);
    } else {
      lds_dma_16_at<i * 1024>(ts.rsrc, v_lds, vrel[i], 0u);
    }
  };

  // the same pieces riding on an MFMA (kind: 0 first of a chain, 1 accumulate in VGPRs, 2 accumulate in an AGPR tile); the row-addressed form keeps
  // its own statement
  constexpr bool kFuse = !kRowDma;  // DMA pieces ride on the MFMA in front of them (the scalar row form of the D = 512 bias / mask / dropout builds issues its own pieces)
  auto issue_k_on = [&](auto ic, int key0, auto kindc, f32x4& d, v8 fa, v8 fb) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    const TileSrc ts = tile_src<BC>(Kg, k_row_bytes, key0, a.Nkv, rb_valid);
    M::template with_dma<decltype(kindc)::value, (kKS && i >= kH ? 2 * PPW + (i - kH) : i) * 1024>(d, fa, fb, ts.rsrc, k_lds, krel[kRowDma ? 0 : i], 0u);
  };
  auto issue_v_on = [&](auto ic, int key0, auto kindc, f32x4& d, v8 fa, v8 fb) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    const TileSrc ts = tile_src<BC>(Vg, v_row_bytes, key0, a.Nkv, rb_valid);
    M::template with_dma<decltype(kindc)::value, i * 1024>(d, fa, fb, ts.rsrc, v_lds, vrel[kRowDma ? 0 : i], 0u);
  };

  // ---- L2 prefetch (FwdArgs::l2_prefetch, set by the launch side for streams that come from HBM): the tile two steps ahead is touched —
  // one dword per 128-byte line — so that the LDS-DMA pieces of the step after next find their lines in the XCD's L2.  The DMA queue covers one
  // step of latency (K(j+1) goes out in step j's softmax phase); a first touch that misses L2 AND the Infinity Cache takes longer than that,
  // and the workgroups that share a K/V stream through one L2 all queue behind the same in-flight lines (measured: 3.4 us per 64-key step at
  // D = 512 while K/V sit in the Infinity Cache, 4.3 us once all of K/V come from HBM).  The workgroups of a head walk K/V together, so each
  // one touches only 1/kPfSlices of a tile: the keys [slice kPfKeys, slice kPfKeys + kPfKeys) with slice = (row tile + head) mod kPfSlices,
  // one 64-lane load for K (wave 0) and one for V (wave 1), issued behind the step's last DMA piece (loads retire in order: the wait at
  // barrier B then leaves exactly this one outstanding, and it has a whole step to land before the next counted wait).
  constexpr int kPfLines = (RB + 127) / 128;                                                            // lines per row
  constexpr int kPfLinesP2 = kPfLines <= 1 ? 1 : kPfLines <= 2 ? 2 : kPfLines <= 4 ? 4 : kPfLines <= 8 ? 8 : 16;
  static_assert(kPfLines <= 16, "head dims up to 1024");
  constexpr int kPfKeys = 64 / kPfLinesP2;                                                              // keys one load covers
  constexpr int kPfSlices = BC / kPfKeys > 0 ? BC / kPfKeys : 1;
  // (built into the split-D tiles only: at D <= 512 a DMA piece has a whole step to land, the touches cost 1 ... 2 %, and the dropout +
  // bias builds there have no register to spare)
  constexpr bool kPf = ND == 2;
  const bool pf_on = kPf && a.l2_prefetch != 0 && wave < FFPA_M16_PF_WAVES && ((FFPA_M16_PF_WHICH >> (wave & 1)) & 1);
  const bool pf_k = (wave & 1) == 0;  // even waves touch K, odd waves V
  uint32_t pf_off = kDmaOob;
  uint32_t pf_junk = 0u;  // (the loads' destination: never read, but live through the loop so that nothing else is allocated to it)
  if (pf_on) {
    constexpr int kPer = FFPA_M16_PF_WAVES / 2;  // slices one workgroup touches per step
    constexpr int kGroups = kPfSlices / kPer > 0 ? kPfSlices / kPer : 1;
    // which slice: by the workgroup's sequence number ON ITS XCD (the hardware deals workgroup ids round-robin to the 8 XCDs: id >> 3) — the ~32
    // workgroups resident on an XCD have consecutive numbers, so together they touch every slice of the tile in that XCD's L2.  (Round 3 used
    // (row tile + head) mod slices: with two XCDs per head — config 3 — an XCD holds only the even or only the odd row tiles of a head, i.e. four
    // of the eight slices: half of every tile's lines were never touched in its L2.)
    const int slice_seq = (int)(blockIdx.x >> 3);
    const int slice = ((slice_seq % kGroups) * kPer + (wave >> 1)) % kPfSlices;
    const uint32_t key = (uint32_t)(slice * kPfKeys + lane / kPfLinesP2), line = (uint32_t)(lane % kPfLinesP2);
    if (line * 128u < rb_valid && key < (uint32_t)BC) pf_off = key * (pf_k ? k_row_bytes : v_row_bytes) + line * 128u;
  }
  // (scalar by construction — wave is — and pinned: they feed the descriptor of the asm below)
  const uint64_t pf_base64 = (uint64_t)(pf_k ? (const void*)Kg : (const void*)Vg);
  const void* const pf_base = (const void*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pf_base64 >> 32)) << 32) |
                                            (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pf_base64));
  const uint32_t pf_row_bytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pf_k ? k_row_bytes : v_row_bytes));
  auto issue_prefetch = [&](int key0) {
    const TileSrc ts = tile_src<BC>(pf_base, pf_row_bytes, key0, a.Nkv, rb_valid);
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "+v"(pf_junk) : "v"(pf_off), "s"(ts.rsrc) : "memory");
  };

  // ---- KV tile range
  int nt = (a.Nkv + BC - 1) / BC;
  if (a.causal) {
    const int last_row = a.causal_row_mod ? a.causal_row_mod - 1 : q0 + BR - 1;
    const int64_t last = (int64_t)last_row + a.causal_offset;
    const int ntc = last < 0 ? 0 : (int)(last / BC) + 1;
    nt = nt < ntc ? nt : ntc;
  }
  int t0 = split * a.tiles_per_split;
  {
    const int t1 = t0 + a.tiles_per_split;
    nt = nt < t1 ? nt : t1;
  }
  // mask ranges (ffpa_fwd_params.kv_bounds): KV tiles no row of this row tile can see are skipped; keys [free_lo, free_hi) are
  // visible to EVERY row of this wave's 32-row block — tiles inside that range do not read the mask at all
  int free_lo = 0, free_hi = 0;
  if (MASK && a.kv_bounds != nullptr) {
    const int* bp = a.kv_bounds + b * a.s_bounds[0] + hq * a.s_bounds[1];
    int first = 0x7fffffff, end = 0;
#pragma unroll
    for (int blk = 0; blk < BR / 32; ++blk) {
      const int r32 = q0 / 32 + blk;
      if (r32 * 32 < a.Nq) {
        const int lo = bp[4 * r32], hi = bp[4 * r32 + 1];
        first = first < lo ? first : lo;
        end = end > hi ? end : hi;
      }
    }
    const int tf = first / BC, te = (end + BC - 1) / BC;
    // (wave-uniform by construction; pinned to scalar registers: they feed the tile descriptors of the LDS-DMA asm)
    t0 = __builtin_amdgcn_readfirstlane(t0 > tf ? t0 : tf);
    nt = __builtin_amdgcn_readfirstlane(nt < te ? nt : te);
    const int r32w = q0 / 32 + qb;
    if (r32w * 32 < a.Nq) {
      free_lo = __builtin_amdgcn_readfirstlane(bp[4 * r32w + 2]);
      free_hi = __builtin_amdgcn_readfirstlane(bp[4 * r32w + 3]);
    } else {
      free_hi = 0x7fffffff;  // a row block past the last query row: nothing it computes is stored
    }
  }

  // ---- Q fragments (B operand of S^T): lane (n, c) holds Q[row 16 rh + n][32 s + 8 c .. + 8]
  v8 qf[KS][2];
#pragma unroll
  for (int rh = 0; rh < 2; ++rh) {
    const T* qp = (const T*)a.q + b * a.sq[0] + hq * a.sq[1] + (int64_t)qrow_c[rh] * a.sq[2] + dh * DW + c * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      qf[s][rh] = (dh * DW + s * 32 + c * 8 < a.d_valid && a.q_mode != 1) ? *(const v8*)(qp + s * 32) : __builtin_bit_cast(v8, z);
    }
  }
  if (a.q_mode == 2) {  // a negative softmax scale reaches the kernel as (-Q, |scale|): exact (a sign flip), and the scale below is always > 0
#pragma unroll
    for (int rh = 0; rh < 2; ++rh)
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        u32x4 w = __builtin_bit_cast(u32x4, qf[s][rh]);
        w ^= (u32x4)(0x80008000u);
        qf[s][rh] = __builtin_bit_cast(v8, w);
      }
  }

  f32x4 oacc[NDB][2];
#pragma unroll
  for (int i = 0; i < NDB; ++i) {
    oacc[i][0] = (f32x4)(0.f);
    oacc[i][1] = (f32x4)(0.f);
  }
  float m_run[2] = {-INFINITY, -INFINITY};  // running row max (log2 domain), the same value in the row's 4 lanes
  float l_run[2] = {0.f, 0.f};              // this lane's share of the row sum

  // ---- per-lane fragment addresses (one base per 64-key half of the tile; everything else is an immediate)
  // K fragment of step s = KV q + i, key block kb: kaddr[kb / 4][i] + KVB q + (kb % 4) * 16 * RB: lane (n, c) reads key 16 kb + n,
  // slot (4 s + c) ^ swizzle(key) (the XOR stays inside the low slot bits the variants enumerate)
  FFPA_LDS const char* kaddr[NH][KV];
#pragma unroll
  for (int hf = 0; hf < NH; ++hf)
#pragma unroll
    for (int i = 0; i < KV; ++i) kaddr[hf][i] = Kt + (64 * hf + n16) * RB + (((dh * (DW / 8) + 4 * i + c) ^ m16_k_swizzle<D>(n16)) << 4);
  // V^T fragment of column block db = VV q + i, key step ks: lane L = lane % 16 of group c reads key 32 ks + 4 c + L / 4 (+ 16 for
  // the second read), 4 columns 16 db + 4 (L % 4) ..: vaddr[ks / 2][i] + VVB q + ((ks % 2) * 32 + {0, 16}) * RB
  FFPA_LDS const char* vaddr[NH][VV];
  {
    const int vkey = 4 * c + (n16 >> 2);
    const int sw = m16_v_swizzle<D>(vkey);
#pragma unroll
    for (int hf = 0; hf < NH; ++hf)
#pragma unroll
      for (int i = 0; i < VV; ++i)
        vaddr[hf][i] = Vt + (64 * hf + vkey) * RB + (((dh * (DW / 8) + 2 * i + ((n16 & 3) >> 1)) ^ sw) << 4) + 8 * (n16 & 1);
  }

  // ---- additive bias (MK == 1): where the initial S^T accumulators of a KV step come from (see the header)
  //   a.bias_lds > 0: key bias, fp32 / scale row cache in LDS (this many bytes: a whole number of tiles of fp32);
  //   a.bias_tile:    [32 rows x BC keys] tiles of the caller's dtype staged by LDS-DMA one step ahead, a.bias_lds = -(bytes of all staging areas);
  //   else:           element-wise global loads.
  // D > 512 (two waves per row block, partial S^T tiles summed): the dh = 0 wave alone carries the bias.
  const bool bias_owner = kBias && dh == 0 && a.bias_dtype != 0;
  const int b_esz = a.bias_dtype == 3 ? 4 : 2;        // staged dtypes: fp16 / bf16 / fp32
  const int b_rowb = BC * b_esz;                      // bytes of one staged row: 64 ... 256
  const int b_slots = b_rowb >> 4;                    // 16-byte slots per staged row: 4, 8 or 16
  const int b_sh = b_slots == 16 ? 0 : (b_slots == 8 ? 1 : 2);  // staged rows per 256-byte bank row = 1 << b_sh
  const int b_pieces = (MK == 1 && a.bias_tile && bias_owner) ? (b_slots >> 1) : 0;  // 1 KiB pieces per [32 rows x BC keys] tile: 2, 4 or 8
  constexpr int kBtMax = BC == 64 ? 8 : 4;            // ... at most (fp32)
  FFPA_LDS char* const Bt = Bl + qb * (32 * b_rowb);  // this wave's staging area
  const uint32_t bt_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)Bt);
  // staged image: row-major [32][BC], 16-byte slot s of row r stored at slot s ^ g(r), g(r) = (r >> b_sh) & (b_slots - 1) (source-side swizzle:
  // the ds_read_b64 / b128 lane groups below — 16 rows x one or two slots — are conflict-free).  Piece i covers rows i * rpp .. + rpp
  // (rpp = 64 / b_slots); its lane l fetches (row l / b_slots, slot (l % b_slots) ^ g): g = (4 i & (b_slots - 1)) | ((l / b_slots) >> b_sh).
  uint32_t b_rowoff = 0, b_col = 0;
  uint32_t b_piece_rows = 0;          // bytes between the first rows of two consecutive pieces: (64 / b_slots) row strides
  // descriptor of this wave's 32 bias rows, from key 0 on (the step's first key goes into the scalar offset): kept as base + byte count and
  // assembled right in front of each piece, like the K / V tile descriptors (a 4-dword descriptor held live across the kernel gets parked in
  // VGPR lanes under scalar pressure, and an inline-asm "s" operand of vector type is then handed over as VGPRs: an assembler error at best)
  const char* b_base = nullptr;
  uint32_t b_left = 0;
  FFPA_LDS const char* baddr[NKB];  // staged tile: this lane's read address of key block kb, row half 0 (row half 1: + 16 rows)
  if constexpr (MK == 1) {
    if (b_pieces > 0) {
      const uint32_t b_rs = (uint32_t)a.sbias[2] * (uint32_t)b_esz;
      const int r = lane / b_slots, sl = lane % b_slots;
      b_rowoff = (uint32_t)r * b_rs;
      b_col = (uint32_t)((sl ^ (r >> b_sh)) << 4);
      b_piece_rows = (uint32_t)(4 << b_sh) * b_rs;
      // descriptor over [this wave's first row, end of the plane's last row): rows past Nq read as zeros; 32 rows span < 2 GiB (row stride < 2^24
      // elements), so the 32-bit offsets never wrap however large the bias tensor is
      const int64_t plane = (int64_t)b_esz * (b * a.sbias[0] + hq * a.sbias[1]);
      const int64_t first = (int64_t)wq0 * b_rs;
      int64_t left = (int64_t)(a.Nq - 1) * b_rs + (int64_t)b_esz * a.Nkv - first;
      left = left < 0 ? 0 : (left > 0x7fffffff ? 0x7fffffff : left);
      b_base = (const char*)a.bias + plane + first;
      b_left = (uint32_t)left;
      const int g = (n16 >> b_sh) & (b_slots - 1);
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const int slot = b_esz == 4 ? 4 * kb + c : 2 * kb + (c >> 1);
        baddr[kb] = Bt + n16 * b_rowb + ((slot ^ g) << 4) + (b_esz == 4 ? 0 : 8 * (c & 1));
      }
    }
  }
  // piece i of the bias tile of the KV step starting at key0, for this wave's 32 rows
  // (the descriptor is assembled from its scalars once per KV step, right in front of the pieces that use it)
  auto bias_rsrc = [&]() -> u32x4 {
    const uint64_t bb = (uint64_t)b_base;  // (wave-uniform by construction; pinned to scalar registers for the asm's "s" operands)
    const u32x4 r = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bb), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bb >> 32)) & 0xffffu,
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)b_left), 0x00020000u};
    return r;
  };
  auto issue_bias = [&](auto ic, int key0, u32x4 rsrc) {
    if constexpr (MK == 1) {
      constexpr int i = decltype(ic)::value;
      const uint32_t voff = (b_col ^ (uint32_t)((64 * i) & (b_rowb - 1))) + b_rowoff;
      // (wave-uniform values, pinned to scalar registers: M0 and the scalar offset of the DMA asm)
      const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)i * b_piece_rows + (uint32_t)(key0 * b_esz)));
      lds_dma_16_at<i * 1024>(rsrc, bt_s, voff, soff);
    }
  };
  FFPA_LDS const char* const bl_lane = Bl + 16 * c;  // row cache: this lane's 4 keys of block kb of the step at k0 sit at bl_lane + 4 k0 + 64 kb

  if (kBias && a.bias_lds > 0 && nt > t0) {
    // key bias [.., .., 1, Nkv]: every row of the workgroup adds the same Nkv values — converted once to fp32 / scale (entries past Nkv are
    // zeros: those keys get the tail mask); made visible by the barrier below
    const int64_t src0 = b * a.sbias[0] + hq * a.sbias[1];
    if (a.bias_cache_raw) {  // (a key row too long for the fp32 form: the caller's 16-bit elements, converted at the top of every step)
      const int n_ent = a.bias_lds >> 1;
      for (int i = tid; i < n_ent; i += 256) *(FFPA_LDS uint16_t*)(Bl + 2 * i) = i < a.Nkv ? ((const uint16_t*)a.bias)[src0 + i * a.sbias[3]] : (uint16_t)0;
    } else {
      const int n_ent = a.bias_lds >> 2;
      for (int i = tid; i < n_ent; i += 256) {
        float w = 0.f;
        if (i < a.Nkv) {
          const int64_t e = src0 + i * a.sbias[3];
          w = a.bias_dtype == 3 ? ((const float*)a.bias)[e] : a.bias_dtype == 2 ? (float)((const __bf16*)a.bias)[e] : (float)((const _Float16*)a.bias)[e];
          w *= a.inv_scale;
        }
        *(FFPA_LDS float*)(Bl + 4 * i) = w;
      }
    }
  }
  if (nt > t0) {
    static_for<PPW>([&](auto ic) { issue_k(ic, t0 * BC); });
    if constexpr (MK == 1) {
      const u32x4 brs = bias_rsrc();
      static_for<kBtMax>([&](auto ic) {
        if (decltype(ic)::value < b_pieces) issue_bias(ic, t0 * BC, brs);
      });
    }
    dma_wait_all();
    __syncthreads();
  }

#ifdef FFPA_M16_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
  const unsigned long long tstart = tprev;
#endif
  if constexpr (kPipe) {
    // =====================================================================================================================================
    // The softmax pipeline of the split-D tiles (header comment above the FFPA_M16_PP_* piece counts).  Same arithmetic as the loop below, instruction
    // for instruction per score — only WHEN each part runs differs: bit-identical outputs (tests/test_m16_gpu.py).
    // =====================================================================================================================================
    constexpr int kXW = 6144;  // exchange bytes per wave: block 0 of even tiles, block 0 of odd tiles, block 1
    constexpr int PFS = FFPA_M16_PP_PF;
    auto k_frag_sk = [&](int s_, int kb_) -> v8 { return *(FFPA_LDS const v8*)(kaddr[kb_ / 4][s_ % KV] + (s_ / KV) * KVB + (kb_ % 4) * 16 * RB); };
    FFPA_LDS char* const xw = Xb + wave * kXW + lane * 16;        // this wave's partials
    FFPA_LDS const char* const xr = Xb + (wave ^ 1) * kXW + lane * 16;  // the other D-half's
    // the softmax shared by rows (see the S phase): where a wave's P^T fragment (lane x 16 B) and rescale factor (lane x 4 B) travel.
    //   builds with LDS to spare (MK 0 / 2): an area of their own behind the partials — P^T [row block][row half] x 1 KiB, scalars [row block][row half] x 256 B;
    //   the key-bias build (the ring takes what is left): the slots of the wave's OWN partials that only it reads back — (key block 1, row half dh) and (key
    //   block 0 of this tile's parity, row half dh): both have been read by the time the softmax ends, the partner reads them behind barrier A2, and the partials of
    //   the next tile overwrite them behind barrier B at the earliest; row half rh's slots sit in the area of the row block's wave dh = rh.
    constexpr bool kOwnSlots = MK == 3;
    FFPA_LDS char* const Pb = Xb + 4 * kXW;
    FFPA_LDS char* const Ab = Pb + 4 * 1024;
    FFPA_LDS char* const pxw = kOwnSlots ? Xb + wave * kXW + 4096 + dh * 1024 + lane * 16 : Pb + (qb * 2 + dh) * 1024 + lane * 16;
    FFPA_LDS const char* const pxr0 = kOwnSlots ? Xb + (qb * 2) * kXW + 4096 + lane * 16 : Pb + qb * 2048 + lane * 16;
    FFPA_LDS const char* const pxr1 = kOwnSlots ? Xb + (qb * 2 + 1) * kXW + 4096 + 1024 + lane * 16 : Pb + qb * 2048 + 1024 + lane * 16;
    constexpr int kAPar = kOwnSlots ? 2048 : 0;  // own slots: the scalar sits in the key-block-0 slot of the tile's parity
    FFPA_LDS char* const axw_base = kOwnSlots ? Xb + wave * kXW + dh * 1024 + lane * 16 : Ab + (qb * 2 + dh) * 256 + lane * 4;
    FFPA_LDS const char* const axr0_base = kOwnSlots ? Xb + (qb * 2) * kXW + lane * 16 : Ab + qb * 512 + lane * 4;
    FFPA_LDS const char* const axr1_base = kOwnSlots ? Xb + (qb * 2 + 1) * kXW + 1024 + lane * 16 : Ab + qb * 512 + 256 + lane * 4;
    const int qrow_own = wq0 + 16 * dh + n16;  // the row of this lane in the half whose softmax this wave runs
    const int qrow_c_own = qrow_own < a.Nq ? qrow_own : a.Nq - 1;
    float m_own = -INFINITY, l_own = 0.f;      // its running max (log2 domain) and this lane's share of its row sum
    // Key-bias build (MK 3) on the pipeline, round 5: the bias enters through the S^T accumulators as in the loop below (the dh = 0 wave's chains start from
    // bias / scale, the other's from zero: bit-identical to that loop), read from a RING of FwdArgs.bias_lds / 4 fp32 entries (2048: all the LDS has left next to the
    // pipeline's exchange) indexed by key mod size: filled with keys [0, size) in front of the loop and refilled half a ring at a time, 1024 keys ahead of the walk.
    const uint32_t ring_mask = MK == 3 ? (uint32_t)((a.bias_lds >> 2) - 1) : 0u;
    auto bias_init = [&](f32x4 (&s)[2], int key0) __attribute__((always_inline)) {
      if constexpr (MK == 3) {
        // no branch in the MFMA stream: both waves read the quad, the wave that does not carry the bias selects zeros (a select, not a multiply by 0: a
        // key-padding bias holds -inf)
        const f32x4 w = *(FFPA_LDS const f32x4*)(Bl + ((((uint32_t)key0 + 4u * (uint32_t)c) & ring_mask) << 2));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[0][r] = bias_owner ? w[r] : 0.f;
          s[1][r] = bias_owner ? w[r] : 0.f;
        }
        asm volatile("s_nop 1" : "+v"(s[0]), "+v"(s[1]));  // VALU write -> MFMA SrcC read wait states (the MFMAs are inline asm)
      }
    };
    constexpr bool kFromBias = MK == 3;  // the chains accumulate onto their initial value instead of starting from zero
    if (nt > t0) {
      f32x4 s0[2];  // partial S^T of key block 0 of the first tile (this wave's D-half); in the loop it is contracted one step early
      // prologue: key block 0 of the first tile (K(t0) has landed and is visible: the barrier above)
      v8 kf[KS];
#pragma unroll
      for (int n = 0; n < PF1 && n < KS; ++n) kf[n] = k_frag_sk(n, 0);
      bias_init(s0, t0 * BC);
      static_for<KS>([&](auto sc) {
        constexpr int s_ = decltype(sc)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (s_ + PF1 < KS) kf[s_ + PF1] = k_frag_sk(s_ + PF1, 0);
        static_assert(KS >= 2, "the last MFMA of a chain carries the wait states");
        if constexpr (s_ == 0 && !kFromBias) {
          M::first(s0[0], kf[s_], qf[s_][0]);
          M::first(s0[1], kf[s_], qf[s_][1]);
        } else {
          M::acc(s0[0], kf[s_], qf[s_][0]);
          if constexpr (s_ == KS - 1) M::acc_last(s0[1], s0[0], kf[s_], qf[s_][1]);
          else M::acc(s0[1], kf[s_], qf[s_][1]);
        }
      });
      __builtin_amdgcn_sched_barrier(0);
      *(FFPA_LDS f32x4*)(xw + (t0 & 1) * 2048) = s0[0];
      *(FFPA_LDS f32x4*)(xw + (t0 & 1) * 2048 + 1024) = s0[1];
      __syncthreads();  // every wave is done reading keys 0 .. 15 of K(t0): their rows take K1(t0 + 1)
      static_for<ppK1P>([&](auto ic) { issue_k(ic, (t0 + 1) * BC); });
      dma_wait_all();  // (once per workgroup: the counted waits of the loop assume a whole step's pieces behind these)
    }
    for (int j = t0; j < nt; ++j) {
      const int k0 = j * BC;
      if constexpr (MK == 3) {
        // the ring's half that the walk has left behind takes the keys 1024 ... 2048 ahead (nobody reads them before 31 more steps — every barrier
        // in between publishes them —, nobody still reads what they replace: keys below k0).  The loads are the compiler's: its wait in front of the
        // LDS stores also drains the DMA queue, once per 32 steps.
        const int half = (a.bias_lds >> 3);  // entries of half a ring
        if (k0 > t0 * BC && ((k0 - t0 * BC) & (half - 1)) == 0) {
          const int64_t src0 = b * a.sbias[0] + hq * a.sbias[1];
          for (int i = tid; i < half; i += 256) {
            const int key = k0 + half + i;
            float w = 0.f;
            if (key < a.Nkv) {
              const int64_t e = src0 + key * a.sbias[3];
              w = a.bias_dtype == 3 ? ((const float*)a.bias)[e] : a.bias_dtype == 2 ? (float)((const __bf16*)a.bias)[e] : (float)((const _Float16*)a.bias)[e];
              w *= a.inv_scale;
            }
            *(FFPA_LDS float*)(Bl + (((uint32_t)key & ring_mask) << 2)) = w;
          }
        }
      }
      // ================= Q: S^T key block 1 of tile j =================
      f32x4 s1[2];
      {
        v8 kf[KS];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < PF1 && n < KS; ++n) kf[n] = k_frag_sk(n, 1);
        bias_init(s1, k0 + 16);
        static_for<KS>([&](auto sc) {
          constexpr int s_ = decltype(sc)::value;
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (s_ + PF1 < KS) kf[s_ + PF1] = k_frag_sk(s_ + PF1, 1);
          constexpr int t = m16_piece_at(s_, KS, ppK1Q + ppVQ, FFPA_M16_PP_QSTEP);
          if constexpr (t >= 0 && kFuse) {
            using kind = std::integral_constant<int, (s_ == 0 && !kFromBias) ? 0 : 1>;
            if constexpr (t < ppK1Q) issue_k_on(std::integral_constant<int, ppK1P + t>{}, k0 + BC, kind{}, s1[0], kf[s_], qf[s_][0]);
            else issue_v_on(std::integral_constant<int, t - ppK1Q>{}, k0, kind{}, s1[0], kf[s_], qf[s_][0]);
          } else {
            if constexpr (s_ == 0 && !kFromBias) M::first(s1[0], kf[s_], qf[s_][0]);
            else M::acc(s1[0], kf[s_], qf[s_][0]);
            if constexpr (t >= 0) {
              if constexpr (t < ppK1Q) issue_k(std::integral_constant<int, ppK1P + t>{}, k0 + BC);  // K1(j+1), rest
              else issue_v(std::integral_constant<int, t - ppK1Q>{}, k0);                           // V(j), first part
            }
          }
          if constexpr (s_ == 0 && !kFromBias) M::first(s1[1], kf[s_], qf[s_][1]);
          else if constexpr (s_ == KS - 1) M::acc_last(s1[1], s1[0], kf[s_], qf[s_][1]);
          else M::acc(s1[1], kf[s_], qf[s_][1]);
        });
        __builtin_amdgcn_sched_barrier(0);
        *(FFPA_LDS f32x4*)(xw + 4096) = s1[0];
        *(FFPA_LDS f32x4*)(xw + 4096 + 1024) = s1[1];
      }
      FFPA_TSTAMP(0);  // QK^T key block 1 (+ partial S^T stores)
      // barrier A1: partial S^T of block 1 visible; every wave done reading K2(j); K1(j+1) has landed everywhere
      if constexpr (ppK1Q > 0) {
        dma_wait_except<ppVQ>();
      } else {
        if (pf_on) dma_wait_except<ppVQ + 1>();
        else dma_wait_except<ppVQ>();
      }
      __syncthreads();
      FFPA_TSTAMP(1);  // K1(j+1) drain + wait at barrier A1

      // ================= S: softmax of tile j in the gaps of S^T key block 0 of tile j + 1 =================
      // Round 5: the softmax is SHARED BY ROWS between the two waves of a row block.  Both hold the same 32 x 32 scores once the partials are summed, and
      // until round 4 both ran the whole softmax on them (the price of splitting D over waves: twice the exponentials per FLOP).  Now wave (qb, dh) sums and
      // exponentiates only row half dh (rows 16 dh .. + 16 of the block: 8 scores per lane instead of 16, two partner partials to read instead of four) and
      // hands the other wave its P^T fragment — already in the PV MFMA's B-operand layout, 16 bytes per lane — and its rescale factor through LDS, next to
      // barrier A2 which both need anyway.  Same sums in the same order per score and per row (a + b == b + a): bit-identical outputs.
      // (this wave's own partials of tile j come back from LDS like the other D-half's: carried in registers across the step, hipcc keeps two sets and
      // copies one into the other at the loop's end — a VALU read of an MFMA result placed inside its wait states)
      f32x4 s0[2];  // partial S^T of key block 0 of tile j + 1
      float x[NKB][4];
      float tmax = 0.f, m_use = 0.f, psum = 0.f, alpha_own = 1.f, earg[2] = {0.f, 0.f};
      v8 pf_own;
      static_assert(NKS == 1, "one P^T fragment per row half and tile");
      {
        f32x4 tp[NKB], xc[NKB];
        __builtin_amdgcn_sched_barrier(0);
        xc[0] = *(FFPA_LDS const f32x4*)(xw + (j & 1) * 2048 + dh * 1024);
        xc[1] = *(FFPA_LDS const f32x4*)(xw + 4096 + dh * 1024);
        tp[0] = *(FFPA_LDS const f32x4*)(xr + (j & 1) * 2048 + dh * 1024);
        tp[1] = *(FFPA_LDS const f32x4*)(xr + 4096 + dh * 1024);
        v8 kf[KS];
#pragma unroll
        for (int n = 0; n < PFS && n < KS; ++n) kf[n] = k_frag_sk(n, 0);
        bias_init(s0, k0 + BC);
        __builtin_amdgcn_sched_barrier(0);
        const bool tail = k0 + BC > a.Nkv;
        const bool diag = a.causal && ((int64_t)k0 + BC - 1 > (int64_t)(a.causal_row_mod ? 0 : wq0) + a.causal_offset);
        // one group of the softmax's instructions per MFMA gap: g = 2 * fragment + (0: behind the first, 1: behind the second MFMA)
        auto softmax_gap = [&](auto gc) __attribute__((always_inline)) {
          constexpr int g = decltype(gc)::value;
          if constexpr (g == 3 || g == 5) {  // + the other D-half's partial
            constexpr int kb = (g - 3) >> 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) x[kb][r] = xc[kb][r] + tp[kb][r];
          } else if constexpr (g == 7) {
            if constexpr (MK == 2) {
              // boolean mask bytes (non-zero = visible), straight from the caller's tensor, exactly as the loop below reads them; steps in the mask's
              // neutral interior (kv_bounds) read nothing.  (The loads are the compiler's: its wait in front of their first use also drains the DMA
              // pieces issued so far in this phase — only on the steps that read the mask.)
              const bool mask_free = k0 >= free_lo && k0 + BC <= free_hi;
              if (a.bias_dtype == 4 && !mask_free) {
                const uint8_t* mr = (const uint8_t*)a.bias + b * a.sbias[0] + hq * a.sbias[1] + (int64_t)qrow_c_own * a.sbias[2];
                if (a.bias_vec == 16 && k0 + BC <= a.Nkv) {
                  uint32_t raw[NKB];
#pragma unroll
                  for (int kb = 0; kb < NKB; ++kb) raw[kb] = *(const uint32_t*)(mr + k0 + kb * 16 + 4 * c);
#pragma unroll
                  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                      if (((raw[kb] >> (8 * r)) & 0xffu) == 0u) x[kb][r] = -INFINITY;
                } else {
#pragma unroll
                  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                      int key = k0 + kb * 16 + 4 * c + r;
                      key = key < a.Nkv ? key : a.Nkv - 1;
                      if (mr[key * a.sbias[3]] == 0) x[kb][r] = -INFINITY;
                    }
                }
              }
            }
            if (tail || diag) {
              const int crow = a.causal_row_mod ? qrow_own % a.causal_row_mod : qrow_own;
              const int64_t lim = a.causal ? (int64_t)crow + a.causal_offset : (int64_t)a.Nkv;
#pragma unroll
              for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const int key = k0 + kb * 16 + 4 * c + r;
                  if (key >= a.Nkv || key > lim) x[kb][r] = -INFINITY;
                }
            }
          } else if constexpr (g == 8) {  // row max, this lane's 8 keys (same order as the loop below)
            float t = x[0][0];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
              for (int r = 0; r < 4; ++r) t = fmaxf(t, x[kb][r]);
            tmax = t;
          } else if constexpr (g == 10) {
            row4_reduce1<true>(tmax);
            tmax *= a.scale_log2;
          } else if constexpr (g == 12) {
            // lazy rescale (threshold FwdArgs.thr): the factor goes to BOTH waves of the row block (each owns half of O^T's columns of these rows) next to
            // the P^T fragment; O^T itself is scaled behind barrier A2, where the partner's factor is known too
            const float m_new = fmaxf(m_own, tmax);
            const bool grow = m_new > m_own + a.thr;
            alpha_own = grow ? __builtin_amdgcn_exp2f(m_own - m_new) : 1.f;
            l_own *= alpha_own;
            m_own = grow ? m_new : m_own;
            m_use = (m_own == -INFINITY) ? 0.f : m_own;
          } else if constexpr (g >= 14 && g <= 28 && (g & 1) == 0) {  // one exponential every other gap, in the loop's order (the row sum adds up in that order)
            constexpr int i = (g - 14) >> 1, kb = (i >> 2) & 1, r = i & 3;
            if constexpr ((r & 1) == 0) {  // (the pair's two exponents in one packed FMA: the same roundings)
              typedef __attribute__((ext_vector_type(2))) float f32x2;
              const f32x2 xv = {x[kb][r], x[kb][r + 1]};
              const f32x2 av = __builtin_elementwise_fma(xv, (f32x2)(a.scale_log2), (f32x2)(-m_use));
              earg[0] = av[0];
              earg[1] = av[1];
            }
            const float pv = __builtin_amdgcn_exp2f(earg[r & 1]);
            psum += pv;
            pf_own[4 * kb + r] = (T)pv;
          } else if constexpr (g == 30) {
            l_own += psum;
          }
        };
        // the 32 groups are written for the 32 MFMA gaps of D = 1024; a smaller head dim has 2 KS < 32 gaps and takes several groups per gap
        auto softmax_gaps = [&](auto gc) __attribute__((always_inline)) {
          constexpr int g = decltype(gc)::value;
          constexpr int lo = (g * 32 + 2 * KS - 1) / (2 * KS), hi = ((g + 1) * 32 + 2 * KS - 1) / (2 * KS);
          static_for<hi - lo>([&](auto uc) { softmax_gap(std::integral_constant<int, lo + decltype(uc)::value>{}); });
        };
        static_for<KS>([&](auto sc) {
          constexpr int s_ = decltype(sc)::value;
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (s_ + PFS < KS) kf[s_ + PFS] = k_frag_sk(s_ + PFS, 0);
          constexpr int t = m16_piece_at(s_, KS, ppVS + ppK2S, FFPA_M16_PP_SSTEP);
          if constexpr (t >= 0 && kFuse) {
            using kind = std::integral_constant<int, (s_ == 0 && !kFromBias) ? 0 : 1>;
            if constexpr (t < ppVS) issue_v_on(std::integral_constant<int, ppVQ + t>{}, k0, kind{}, s0[0], kf[s_], qf[s_][0]);
            else issue_k_on(std::integral_constant<int, kH + (t - ppVS)>{}, k0 + BC, kind{}, s0[0], kf[s_], qf[s_][0]);
          } else {
            if constexpr (s_ == 0 && !kFromBias) M::first(s0[0], kf[s_], qf[s_][0]);
            else M::acc(s0[0], kf[s_], qf[s_][0]);
            if constexpr (t >= 0) {
              if constexpr (t < ppVS) issue_v(std::integral_constant<int, ppVQ + t>{}, k0);  // V(j), rest
              else issue_k(std::integral_constant<int, kH + (t - ppVS)>{}, k0 + BC);        // K2(j+1), first part
            }
          }
          constexpr bool kPerGap = KS >= 16 && MK != 2;  // one group per MFMA gap at D = 1024, all behind the phase's MFMAs elsewhere (see the header of the pipeline)
          if constexpr (kPerGap) softmax_gaps(std::integral_constant<int, 2 * s_>{});
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (s_ == 0 && !kFromBias) M::first(s0[1], kf[s_], qf[s_][1]);
          else if constexpr (s_ == KS - 1) M::acc_last(s0[1], s0[0], kf[s_], qf[s_][1]);
          else M::acc(s0[1], kf[s_], qf[s_][1]);
          if constexpr (kPerGap) softmax_gaps(std::integral_constant<int, 2 * s_ + 1>{});
        });
        if constexpr (KS < 16 || MK == 2) static_for<2 * KS>([&](auto gc) { softmax_gaps(gc); });
        __builtin_amdgcn_sched_barrier(0);
        *(FFPA_LDS f32x4*)(xw + ((j + 1) & 1) * 2048) = s0[0];
        *(FFPA_LDS f32x4*)(xw + ((j + 1) & 1) * 2048 + 1024) = s0[1];
        // this wave's share of the softmax for the other D-half's wave (read behind barrier A2; the slot is free: its last readers passed barrier B)
        *(FFPA_LDS v8*)(pxw) = pf_own;
        *(FFPA_LDS float*)(axw_base + (j & 1) * kAPar) = alpha_own;
      }

      // ================= P: O^T += V^T.P^T =================
      {
        __builtin_amdgcn_sched_barrier(0);
        FFPA_TSTAMP(2);  // softmax(j) + QK^T(j+1) key block 0
        // barrier A2: V(j) has landed on every wave; every wave is done reading K1(j+1)'s rows ... no: done CONTRACTING them — they take K1(j+2)
        dma_wait_except<ppWaitA2>();
        __syncthreads();
        FFPA_TSTAMP(3);  // V(j) drain + wait at barrier A2
        // both row halves' P^T fragments and rescale factors (this wave's own come back from LDS too: no register selects on the wave's D-half index)
        v8 pf[NKS][2];
        pf[0][0] = *(FFPA_LDS const v8*)(pxr0);
        pf[0][1] = *(FFPA_LDS const v8*)(pxr1);
        const float alpha0 = *(FFPA_LDS const float*)(axr0_base + (j & 1) * kAPar), alpha1 = *(FFPA_LDS const float*)(axr1_base + (j & 1) * kAPar);
        v8 vf[N2];
        auto v_frag = [&](int n) -> v8 {
          const int db = n % NDB, ks = n / NDB;
          FFPA_LDS const char* vp = vaddr[ks / 2][db % VV] + (db / VV) * VVB + (ks % 2) * 32 * RB;
          const v4 lo = E::tr_read(vp);
          const v4 hi = E::tr_read(vp + 16 * RB);
          return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        };
#pragma unroll
        for (int n = 0; n < PF2 && n < N2; ++n) vf[n] = v_frag(n);
        if (j > t0 && __any(alpha0 != 1.f || alpha1 != 1.f)) {
          // rare path: a row's max grew by more than the threshold — O^T (AGPRs) is scaled in place through one temporary VGPR tile
#pragma unroll
          for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
              f32x4 t = oacc[i][rh];
              asm volatile("" : "+a"(t));
              t *= (rh ? alpha1 : alpha0);
              asm volatile("" : "+a"(t));
              oacc[i][rh] = t;
              __builtin_amdgcn_sched_barrier(0);
            }
        }
        static_for<N2>([&](auto ic) {
          constexpr int n = decltype(ic)::value;
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (n + PF2 < N2) vf[n + PF2] = v_frag(n + PF2);
          constexpr int t = m16_piece_at(n, N2, ppK2P + ppK1P, FFPA_M16_PP_PSTEP);
          constexpr int db = n % NDB, ks = n / NDB;
          if constexpr (t >= 0 && kFuse) {
            using kind = std::integral_constant<int, 2>;
            if constexpr (t < ppK2P) issue_k_on(std::integral_constant<int, kH + ppK2S + t>{}, k0 + BC, kind{}, oacc[db][0], vf[n], pf[ks][0]);
            else issue_k_on(std::integral_constant<int, t - ppK2P>{}, k0 + 2 * BC, kind{}, oacc[db][0], vf[n], pf[ks][0]);
          } else {
            M::acc_a(oacc[db][0], vf[n], pf[ks][0]);
            if constexpr (t >= 0) {
              if constexpr (t < ppK2P) issue_k(std::integral_constant<int, kH + ppK2S + t>{}, k0 + BC);  // K2(j+1), rest
              else issue_k(std::integral_constant<int, t - ppK2P>{}, k0 + 2 * BC);                       // K1(j+2), first part
            }
          }
          M::acc_a(oacc[db][1], vf[n], pf[ks][1]);
        });
        __builtin_amdgcn_sched_barrier(0);
      }
      FFPA_TSTAMP(4);  // PV loop
      // barrier B: every wave is done reading V(j); K2(j+1) has landed and is visible (the K1(j+2) pieces behind it stay in flight)
      if (pf_on) {  // (wave-uniform)
        issue_prefetch(k0 + FFPA_M16_PF_DIST * BC);
        dma_wait_except<ppWaitB + 1>();
      } else {
        dma_wait_except<ppWaitB>();
      }
      __syncthreads();
      FFPA_TSTAMP(5);  // K2(j+1) drain + wait at barrier B
    }
    // the epilogue below wants both row halves' running max and this lane's share of both row sums: the other half's come from its owner (the
    // exchange slots are free: their last readers passed barrier B of the last step)
    *(FFPA_LDS float*)(axw_base) = m_own;
    *(FFPA_LDS float*)(pxw) = l_own;
    __syncthreads();
    m_run[0] = *(FFPA_LDS const float*)(axr0_base);
    m_run[1] = *(FFPA_LDS const float*)(axr1_base);
    l_run[0] = *(FFPA_LDS const float*)(pxr0);
    l_run[1] = *(FFPA_LDS const float*)(pxr1);
  } else {
  // The DROPOUT builds of the split-D tiles (ND == 2) share the softmax BY ROWS between the two waves of a row block, as the pipelined loop above does: both hold
  // the same 32 x BC scores once the partials are summed; wave (qb, dh) exponentiates — and draws the Philox bits of — row half dh only and trades its P^T fragment
  // and rescale factor through LDS around barrier A2, in the slots of its own partial-S area that nobody else reads.  Measured (profiles/r05_row_shared_softmax.txt):
  // dropout at D = 1024 + 14 % (the Philox rounds halve); WITHOUT dropout the extra LDS round trip in front of the PV loop costs this un-pipelined loop more than
  // half a softmax returns (key bias - 1 %, bias tiles - 2 %, D = 576 ... 960 - 2.5 ... - 8.6 %): those builds keep the softmax in both waves.
  // RHS = row halves whose softmax this wave runs; when shared, entry 0 of the per-row-half arrays below is THE OWN half (dh), and m_run / l_run hold its state in
  // entry 0 until the exchange behind the loop.
  constexpr bool kRowShare = ND == 2 && DROP;
  constexpr int RHS = kRowShare ? 1 : 2;
  int qrow_s[RHS], qrow_cs[RHS];
#pragma unroll
  for (int rh = 0; rh < RHS; ++rh) {
    qrow_s[rh] = kRowShare ? wq0 + 16 * dh + n16 : qrow[rh];
    qrow_cs[rh] = qrow_s[rh] < a.Nq ? qrow_s[rh] : a.Nq - 1;
  }
  FFPA_LDS char* const xw_nd2 = Xb + wave * 4096 + lane * 16;                 // this wave's partial-S area: slot (kb, rh) at + (2 kb + rh) KiB
  FFPA_LDS const char* const xr_nd2 = Xb + (wave ^ 1) * 4096 + lane * 16;     // the other D-half's
  FFPA_LDS const char* const xb_nd2 = Xb + (qb * 2) * 4096 + lane * 16;       // the row block's two areas (wave dh = 0 first)
  for (int j = t0; j < nt; ++j) {
    const int k0 = j * BC;


    // ================= S^T = K.Q^T =================
    f32x4 sacc[NKB][2];
    {
      v8 kf[N1];
      auto k_frag = [&](int n) -> v8 {
        const int s = n / NKB, kb = n % NKB;
        return *(FFPA_LDS const v8*)(kaddr[kb / 4][s % KV] + (s / KV) * KVB + (kb % 4) * 16 * RB);
      };
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < PF1 && n < N1; ++n) kf[n] = k_frag(n);
      __builtin_amdgcn_sched_barrier(0);  // (the first K fragments are on their way while the bias tile below is read and converted)
      if constexpr (kBias) {
        // the accumulators start from bias / softmax_scale (zeros where there is no bias): see the header
        // (mask_free, wave-uniform: this step lies in the neutral interior of the caller's mask (kv_bounds): nothing to read, nothing to add)
        const bool mask_free = MK == 1 && k0 >= free_lo && k0 + BC <= free_hi;
        if (bias_owner && !mask_free) {
          if ((MK == 3 || a.bias_lds > 0) && a.bias_cache_raw) {  // key bias, 16-bit row cache: 4 keys = one ds_read_b64, converted here
            typedef __attribute__((ext_vector_type(4))) __bf16 b4;
            typedef __attribute__((ext_vector_type(4))) _Float16 h4;
            FFPA_LDS const char* bp = Bl + 8 * c + 2 * k0;
  #pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
              f32x4 w;
              if (a.bias_dtype == 2) {
                const b4 t = *(FFPA_LDS const b4*)(bp + 32 * kb);
  #pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = (float)t[r] * a.inv_scale;
              } else {
                const h4 t = *(FFPA_LDS const h4*)(bp + 32 * kb);
  #pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = (float)t[r] * a.inv_scale;
              }
              sacc[kb][0] = w;
              sacc[kb][1] = w;
            }
          } else if (MK == 3 || a.bias_lds > 0) {  // key bias: fp32 / scale from the LDS row cache, the same 4 keys for both of the lane's rows
            FFPA_LDS const char* bp = bl_lane + 4 * k0;
            FFPA_LDS const char* bp2 = bp;
            asm volatile("" : "+v"(bp2));  // (two reads, no register copies: the LDS has the bandwidth, the VALU slots are what the softmax needs)
  #pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
              sacc[kb][0] = *(FFPA_LDS const f32x4*)(bp + 64 * kb);
              sacc[kb][1] = *(FFPA_LDS const f32x4*)(bp2 + 64 * kb);
            }
          } else if (MK == 1 && a.bias_tile) {  // the tile this wave staged during the previous step's PV loop (drained at barrier B)
            typedef __attribute__((ext_vector_type(4))) __bf16 b4;
            typedef __attribute__((ext_vector_type(4))) _Float16 h4;
            const int half_rows = 16 * b_rowb;
            if (a.bias_dtype == 3) {
  #pragma unroll
              for (int kb = 0; kb < NKB; ++kb)
  #pragma unroll
                for (int rh = 0; rh < 2; ++rh) sacc[kb][rh] = *(FFPA_LDS const f32x4*)(baddr[kb] + rh * half_rows) * a.inv_scale;
            } else if (a.bias_dtype == 2) {
  #pragma unroll
              for (int kb = 0; kb < NKB; ++kb)
  #pragma unroll
                for (int rh = 0; rh < 2; ++rh) {
                  const b4 w = *(FFPA_LDS const b4*)(baddr[kb] + rh * half_rows);
  #pragma unroll
                  for (int r = 0; r < 4; ++r) sacc[kb][rh][r] = (float)w[r] * a.inv_scale;
                }
            } else {
  #pragma unroll
              for (int kb = 0; kb < NKB; ++kb)
  #pragma unroll
                for (int rh = 0; rh < 2; ++rh) {
                  const h4 w = *(FFPA_LDS const h4*)(baddr[kb] + rh * half_rows);
  #pragma unroll
                  for (int r = 0; r < 4; ++r) sacc[kb][rh][r] = (float)w[r] * a.inv_scale;
                }
            }
          } else if constexpr (MK == 1) {  // element-wise from global memory: any strides, any dtype (boolean: 0 / -inf)
  #pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
              const int64_t brow = b * a.sbias[0] + hq * a.sbias[1] + (int64_t)qrow_c[rh] * a.sbias[2];
  #pragma unroll
              for (int kb = 0; kb < NKB; ++kb)
  #pragma unroll
                for (int r = 0; r < 4; ++r) {
                  int key = k0 + kb * 16 + 4 * c + r;
                  key = key < a.Nkv ? key : a.Nkv - 1;
                  const int64_t e = brow + key * a.sbias[3];
                  float w;
                  if (a.bias_dtype == 4) w = ((const uint8_t*)a.bias)[e] != 0 ? 0.f : -INFINITY;
                  else w = (a.bias_dtype == 3 ? ((const float*)a.bias)[e] : a.bias_dtype == 2 ? (float)((const __bf16*)a.bias)[e] : (float)((const _Float16*)a.bias)[e]) * a.inv_scale;
                  sacc[kb][rh][r] = w;
                }
            }
          }
        } else {
          // (asm: left to the compiler, these 8 NKB moves are hoisted in front of the branch and paid by the biased steps as well)
  #pragma unroll
          for (int kb = 0; kb < NKB; ++kb)
  #pragma unroll
            for (int rh = 0; rh < 2; ++rh)
              asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0"
                           : "=v"(sacc[kb][rh][0]), "=v"(sacc[kb][rh][1]), "=v"(sacc[kb][rh][2]), "=v"(sacc[kb][rh][3]));
        }
        // VALU write -> MFMA SrcC read wait states (the MFMAs are inline asm); every accumulator is named so that all writes precede the pad
        static_assert(NKB == 2 || NKB == 4, "additive-bias build: 32- or 64-key tiles");
        if constexpr (NKB == 4) asm volatile("" : "+v"(sacc[2][0]), "+v"(sacc[2][1]), "+v"(sacc[3][0]), "+v"(sacc[3][1]));
        asm volatile("s_nop 1" : "+v"(sacc[0][0]), "+v"(sacc[0][1]), "+v"(sacc[1][0]), "+v"(sacc[1][1]));
      }
      static_for<N1>([&](auto ic) {
        constexpr int n = decltype(ic)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (n + PF1 < N1) kf[n + PF1] = k_frag(n + PF1);
        constexpr int s = n / NKB, kb = n % NKB;
        constexpr bool kPiece = n % kStep1 == 0 && n / kStep1 < PPW;
        // (a DMA piece sits BETWEEN the fragment's two MFMAs: + 0.4 ... 1.7 % against in front of / behind them; fused with the first one where the build allows)
        if constexpr (kPiece && kFuse) {
          issue_v_on(std::integral_constant<int, n / kStep1>{}, k0, std::integral_constant<int, (s == 0 && !kBias) ? 0 : 1>{}, sacc[kb][0], kf[n], qf[s][0]);
        } else {
          if constexpr (s == 0 && !kBias) M::first(sacc[kb][0], kf[n], qf[s][0]);
          else M::acc(sacc[kb][0], kf[n], qf[s][0]);
        }
        if constexpr (kPiece && !kFuse) issue_v(std::integral_constant<int, n / kStep1>{}, k0);
        if constexpr (s == 0 && !kBias) M::first(sacc[kb][1], kf[n], qf[s][1]);
        else M::acc(sacc[kb][1], kf[n], qf[s][1]);
      });
      // MFMA result -> VALU reader wait states (invisible to the compiler inside asm); every accumulator is named so that no read
      // of one can be scheduled ahead of the statement
      if constexpr (NKB == 8)
        asm volatile(""
                     : "+v"(sacc[4][0]), "+v"(sacc[4][1]), "+v"(sacc[5][0]), "+v"(sacc[5][1]), "+v"(sacc[6][0]), "+v"(sacc[6][1]), "+v"(sacc[7][0]),
                       "+v"(sacc[7][1]));
      if constexpr (NKB >= 4) asm volatile("" : "+v"(sacc[2][0]), "+v"(sacc[2][1]), "+v"(sacc[3 % NKB][0]), "+v"(sacc[3 % NKB][1]));
      asm volatile("s_nop 15\n\ts_nop 3" : "+v"(sacc[0][0]), "+v"(sacc[0][1]), "+v"(sacc[1][0]), "+v"(sacc[1][1]));
      __builtin_amdgcn_sched_barrier(0);
    }
    static_assert(NKB == 2 || NKB == 4 || NKB == 8, "the wait-state statements above name 4 / 8 / 16 accumulators");

    auto pre_k_group = [&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value;
      if constexpr (kPre >= 4) {
        __builtin_amdgcn_sched_barrier(0);
        static_for<kPre / 4>([&](auto ic) { issue_k(std::integral_constant<int, g * (kPre / 4) + decltype(ic)::value>{}, k0 + BC); });
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    if constexpr (ND == 2) {  // publish this wave's partial S^T (lane-linear, conflict free)
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) *(FFPA_LDS f32x4*)(xw_nd2 + (kb * 2 + rh) * 1024) = sacc[kb][rh];
    }
    FFPA_TSTAMP(0);  // QK^T loop (+ ND == 2: partial S^T stores)
    // barrier A1: every wave is done reading K(j) (ND == 2: and the partial S^T tiles are visible)
    __syncthreads();
    FFPA_TSTAMP(1);  // wait at barrier A1
    __builtin_amdgcn_sched_barrier(0);
    pre_k_group(std::integral_constant<int, 0>{});

    // x[kb][rh][r] = score(row 16 rh + n, key k0 + 16 kb + 4 c + r) / sc, bias included (it entered in units of 1 / sc): the softmax scale
    // is folded into the exponent's FMA (p = exp2(x sc - m): one instruction instead of a multiply here and a subtract there, 32 VALU
    // instructions per tile less) and applied to the row max after its reduction — max(x sc) = sc max(x) for sc > 0, and the kernel only
    // ever sees sc > 0: the launch side turns a negative scale into (-Q, |sc|) and a zero scale into (Q = 0, 1) (FwdArgs.q_mode).
    float x[NKB][RHS][4];
    if constexpr (kRowShare) {
      // the own row half: this wave's partial (back from LDS: no register select on dh) + the other D-half's (a + b == b + a: whichever wave owns a row sees
      // the scores both waves computed until round 4)
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const f32x4 own = *(FFPA_LDS const f32x4*)(xw_nd2 + (kb * 2 + dh) * 1024);
        const f32x4 t = *(FFPA_LDS const f32x4*)(xr_nd2 + (kb * 2 + dh) * 1024);
#pragma unroll
        for (int r = 0; r < 4; ++r) x[kb][0][r] = own[r] + t[r];
      }
    } else if constexpr (ND == 2) {
      // + the other D-half's partial (a + b == b + a: both waves of a row block see bit-identical scores, so their softmax states agree)
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) {
          const f32x4 t = *(FFPA_LDS const f32x4*)(xr_nd2 + (kb * 2 + rh) * 1024);
#pragma unroll
          for (int r = 0; r < 4; ++r) x[kb][rh][r] = sacc[kb][rh][r] + t[r];
        }
    } else {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int rh = 0; rh < 2; ++rh)
#pragma unroll
          for (int r = 0; r < 4; ++r) x[kb][rh][r] = sacc[kb][rh][r];
    }
    pre_k_group(std::integral_constant<int, 1>{});

    if constexpr (MK == 2) {
      // boolean mask bytes (non-zero = visible), straight from the caller's tensor; the lane's 4 keys of a block are consecutive
      const bool mask_free = k0 >= free_lo && k0 + BC <= free_hi;  // wave-uniform: the step lies in the mask's neutral interior (kv_bounds)
      if (a.bias_dtype == 4 && !mask_free) {
        const uint8_t* mp = (const uint8_t*)a.bias + b * a.sbias[0] + hq * a.sbias[1];
#pragma unroll
        for (int rh = 0; rh < RHS; ++rh) {
          const uint8_t* mr = mp + (int64_t)qrow_cs[rh] * a.sbias[2];
          if (a.bias_vec == 16 && k0 + BC <= a.Nkv) {
            uint32_t raw[NKB];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) raw[kb] = *(const uint32_t*)(mr + k0 + kb * 16 + 4 * c);
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (((raw[kb] >> (8 * r)) & 0xffu) == 0u) x[kb][rh][r] = -INFINITY;
          } else {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                int key = k0 + kb * 16 + 4 * c + r;
                key = key < a.Nkv ? key : a.Nkv - 1;
                if (mr[key * a.sbias[3]] == 0) x[kb][rh][r] = -INFINITY;
              }
          }
        }
      }
    }
    const bool tail = k0 + BC > a.Nkv;
    const bool diag = a.causal && ((int64_t)k0 + BC - 1 > (int64_t)(a.causal_row_mod ? 0 : wq0) + a.causal_offset);
    if (tail || diag) {
#pragma unroll
      for (int rh = 0; rh < RHS; ++rh) {
        const int crow = a.causal_row_mod ? qrow_s[rh] % a.causal_row_mod : qrow_s[rh];
        const int64_t lim = a.causal ? (int64_t)crow + a.causal_offset : (int64_t)a.Nkv;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = k0 + kb * 16 + 4 * c + r;
            if (key >= a.Nkv || key > lim) x[kb][rh][r] = -INFINITY;
          }
      }
    }

    // ================= online softmax (prefill.cuh:671-870, log2 domain) =================
    float tmax[RHS];
#pragma unroll
    for (int rh = 0; rh < RHS; ++rh) {
      float t = x[0][rh][0];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) t = fmaxf(t, x[kb][rh][r]);
      tmax[rh] = t;
    }
    if constexpr (RHS == 2) row4_reduce2<true>(tmax[0], tmax[1]);
    else row4_reduce1<true>(tmax[0]);
#pragma unroll
    for (int rh = 0; rh < RHS; ++rh) tmax[rh] *= a.scale_log2;  // (exact: rounding is monotonic, fl(sc max x) = max fl(sc x) for sc > 0; -inf stays -inf)
    pre_k_group(std::integral_constant<int, 2>{});
    float alpha_own = 1.f;  // shared softmax: the own half's rescale factor of this step (1 = none), traded next to the P^T fragment
    if constexpr (kRowShare) {
      const float m_new = fmaxf(m_run[0], tmax[0]);
      const bool grow = m_new > m_run[0] + a.thr;
      alpha_own = grow ? __builtin_amdgcn_exp2f(m_run[0] - m_new) : 1.f;
      l_run[0] *= alpha_own;
      m_run[0] = grow ? m_new : m_run[0];
    }
    const float m_new0 = fmaxf(m_run[0], tmax[0]), m_new1 = fmaxf(m_run[RHS - 1], tmax[RHS - 1]);
    const bool grow0 = !kRowShare && m_new0 > m_run[0] + a.thr, grow1 = !kRowShare && m_new1 > m_run[RHS - 1] + a.thr;
    if (!kRowShare && __any(grow0 || grow1)) {
      const float alpha0 = grow0 ? __builtin_amdgcn_exp2f(m_run[0] - m_new0) : 1.f;
      const float alpha1 = grow1 ? __builtin_amdgcn_exp2f(m_run[RHS - 1] - m_new1) : 1.f;
      if (j > t0) {
        // rare path: O^T lives in AGPRs; scale in place through one temporary VGPR (see ffpa_fwd_kernel.h)
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
          for (int rh = 0; rh < 2; ++rh) {
            f32x4 t = oacc[i][rh];
            asm volatile("" : "+a"(t));
            t *= (rh ? alpha1 : alpha0);
            asm volatile("" : "+a"(t));
            oacc[i][rh] = t;
            __builtin_amdgcn_sched_barrier(0);
          }
      }
      l_run[0] *= alpha0;
      l_run[RHS - 1] *= alpha1;
      m_run[0] = grow0 ? m_new0 : m_run[0];
      m_run[RHS - 1] = grow1 ? m_new1 : m_run[RHS - 1];
    }

    // dropout keep bits of this step (bit 4 kb + r of keep_bits[rh][.] <-> key 16 kb + 4 c + r), drawn BEFORE the exponentials: the Philox
    // temporaries are dead by the time the P^T fragments come alive.  The lane's 4 keys of a block are one Philox group of the row's
    // counter stream (element offset = ((b Hq + hq) Nq + row) Nkv + key).
    uint32_t keep_bits[RHS][NKB > 8 ? NKB / 8 : 1] = {};
    if constexpr (DROP) {
      unsigned long long erow[RHS];
#pragma unroll
      for (int rh = 0; rh < RHS; ++rh)
        erow[rh] = a.philox_offset + (((unsigned long long)b * a.Hq + hq) * a.Nq + (unsigned long long)qrow_cs[rh]) * (unsigned long long)a.Nkv;
      // every lane's 4-key group a whole Philox block (philox_offset and Nkv multiples of 4: the usual case; the key part 16 kb + 4 c always is)?
      // Then the groups of the step are branch-free and sit in ONE basic block.
      if (__builtin_amdgcn_ballot_w64(((erow[0] | erow[RHS - 1]) & 3ull) != 0) == 0ull) {
        constexpr int kIlp = 1;  // Philox groups advanced in lockstep between two scheduling fences (2 / 4 measured: nothing, profiles/r03_philox.txt)
        static_assert(kIlp >= 1 && (RHS * NKB) % kIlp == 0, "Philox groups per batch");
#pragma unroll
        for (int g0 = 0; g0 < RHS * NKB; g0 += kIlp) {
          __builtin_amdgcn_sched_barrier(0);
          unsigned long long quad[kIlp];
          uint32_t bits[kIlp];
#pragma unroll
          for (int i = 0; i < kIlp; ++i) quad[i] = (erow[(g0 + i) % RHS] + (unsigned long long)(k0 + ((g0 + i) / RHS) * 16 + 4 * c)) >> 2;
          dropout_keep_bits4_aligned_n<kIlp, MK == 0>(a.philox_seed, quad, a.keep_threshold, bits);
#pragma unroll
          for (int i = 0; i < kIlp; ++i) keep_bits[(g0 + i) % RHS][((g0 + i) / RHS) >> 3] |= bits[i] << (4 * (((g0 + i) / RHS) & 7));
        }
      } else {
#pragma unroll
        for (int g = 0; g < RHS * NKB; ++g) {
          const int kb = g / RHS, rh = g % RHS;
          __builtin_amdgcn_sched_barrier(0);
          keep_bits[rh][kb >> 3] |= dropout_keep_bits4<MK == 0>(a.philox_seed, erow[rh] + (unsigned long long)(k0 + kb * 16 + 4 * c), a.keep_threshold) << (4 * (kb & 7));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    // P^T fragments: contraction slot 8 c + e of key step ks <-> key 32 ks + 16 (e / 4) + 4 c + e % 4 = x[2 ks + e / 4][rh][e % 4]
    v8 pf[NKS][2];
    v8 pfs[NKS][RHS];
#pragma unroll
    for (int rh = 0; rh < RHS; ++rh) {
      const float m_use = (m_run[rh] == -INFINITY) ? 0.f : m_run[rh];
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // the exponents' x * scale - m as packed FMAs (v_pk_fma_f32: two scores per instruction, the same roundings — bit-identical): config 2 + 0.6 %, causal + 2.0 %,
          // cross + 1.2 %, config 4 + 2.2 %, D = 320 + 0.7 % (interleaved A/B, profiles/r04_pipe.txt)
          float arg;
          // (not in the boolean-mask build of D = 512: there the packed form costs the allocator one scalar lane spill inside the MFMA loops)
          if constexpr (!(D == 512 && MK == 2)) {
            typedef __attribute__((ext_vector_type(2))) float f32x2;
            const f32x2 xv = {x[kb][rh][r & ~1], x[kb][rh][r | 1]};
            const f32x2 av = __builtin_elementwise_fma(xv, (f32x2)(a.scale_log2), (f32x2)(-m_use));
            arg = av[r & 1];
          } else {
            arg = __builtin_fmaf(x[kb][rh][r], a.scale_log2, -m_use);
          }
          const float p = __builtin_amdgcn_exp2f(arg);
          psum += p;  // row sum from the unrounded P (prefill.cuh:755-756)
          if constexpr (DROP) {
            // dropout: applied to the ROUNDED P, after the row sum (LSE is undropped), scaled by 1 / (1 - p) and rounded again
            // (prefill.cuh:508-546); the keep bit was drawn above: AND with 0 / ~0 (P >= 0: no sign games)
            const float scaled = (float)(T)p * a.keep_scale;
            const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)keep_bits[rh][kb >> 3], 4 * (kb & 7) + r, 1);
            pfs[kb >> 1][rh][4 * (kb & 1) + r] = (T)__uint_as_float(__float_as_uint(scaled) & m);
          } else {
            pfs[kb >> 1][rh][4 * (kb & 1) + r] = (T)p;
          }
        }
      l_run[rh] += psum;
    }
    if constexpr (kRowShare) {
      // this wave's share of the softmax for the other D-half's wave, in the two slots of its own partial-S area that only it has read (kb = 0 / 1 of row
      // half dh); read behind barrier A2, overwritten by the next step's partials only behind barrier B
      static_assert(!kRowShare || NKS == 1, "one P^T fragment per row half and tile");
      *(FFPA_LDS v8*)(xw_nd2 + dh * 1024) = pfs[0][0];
      *(FFPA_LDS float*)(xw_nd2 + (2 + dh) * 1024) = alpha_own;
    } else {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int rh = 0; rh < RHS; ++rh) pf[ks][rh] = pfs[ks][rh];
    }
    pre_k_group(std::integral_constant<int, 3>{});

    // ================= O^T += V^T.P^T =================
    {
      __builtin_amdgcn_sched_barrier(0);
      FFPA_TSTAMP(2);  // softmax + the K(j+1) pieces issued inside it
      // barrier A2: V(j) has landed on every wave (all but the kPre younger K pieces have retired)
      dma_wait_except<kPre>();
      __syncthreads();
      FFPA_TSTAMP(3);  // V(j) drain + wait at barrier A2
      if constexpr (kRowShare) {
        // both row halves' P^T fragments and rescale factors: half rh sits in the area of the row block's wave dh = rh, slots (0, rh) and (1, rh)
        pf[0][0] = *(FFPA_LDS const v8*)(xb_nd2);
        pf[0][1] = *(FFPA_LDS const v8*)(xb_nd2 + 4096 + 1024);
        const float alpha0 = *(FFPA_LDS const float*)(xb_nd2 + 2048), alpha1 = *(FFPA_LDS const float*)(xb_nd2 + 4096 + 3072);
        if (j > t0 && __any(alpha0 != 1.f || alpha1 != 1.f)) {
          // rare path: O^T lives in AGPRs; scale in place through one temporary VGPR tile
#pragma unroll
          for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
              f32x4 t = oacc[i][rh];
              asm volatile("" : "+a"(t));
              t *= (rh ? alpha1 : alpha0);
              asm volatile("" : "+a"(t));
              oacc[i][rh] = t;
              __builtin_amdgcn_sched_barrier(0);
            }
        }
      }
        // pieces of the next step's bias tile to stage (none when that step lies in the mask's neutral interior or past the last tile)
      const int b_next = (MK == 1 && j + 1 < nt && !(k0 + BC >= free_lo && k0 + 2 * BC <= free_hi)) ? b_pieces : 0;
      const u32x4 brs = bias_rsrc();
      v8 vf[N2];
      auto v_frag = [&](int n) -> v8 {
        const int db = n % NDB, ks = n / NDB;
        FFPA_LDS const char* vp = vaddr[ks / 2][db % VV] + (db / VV) * VVB + (ks % 2) * 32 * RB;
        const v4 lo = E::tr_read(vp);
        const v4 hi = E::tr_read(vp + 16 * RB);
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      };
#pragma unroll
      for (int n = 0; n < PF2 && n < N2; ++n) vf[n] = v_frag(n);
      static_for<N2>([&](auto ic) {
        constexpr int n = decltype(ic)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (n + PF2 < N2) vf[n + PF2] = v_frag(n + PF2);
        constexpr bool kPiece = n % kStep2 == 0 && n / kStep2 + kPre < PPW;  // the K(j+1) pieces that did not go out between the softmax stages
        constexpr int kIdx = n / kStep2 + kPre;
        constexpr int db = n % NDB, ks = n / NDB;
        if constexpr (kPiece && kFuse) issue_k_on(std::integral_constant<int, kIdx>{}, k0 + BC, std::integral_constant<int, 2>{}, oacc[db][0], vf[n], pf[ks][0]);
        else M::acc_a(oacc[db][0], vf[n], pf[ks][0]);
        if constexpr (kPiece && !kFuse) issue_k(std::integral_constant<int, kIdx>{}, k0 + BC);
        M::acc_a(oacc[db][1], vf[n], pf[ks][1]);
        if constexpr (MK == 1) {
          // the bias tile of step j + 1 (this wave's rows, its private staging area) in the slots the K pieces leave free
          constexpr int kBStep = kStep2 >= 2 ? kStep2 : 2;
          if constexpr (n % kBStep == kBStep / 2 && n / kBStep < kBtMax) {
            if (n / kBStep < b_next) issue_bias(std::integral_constant<int, n / kBStep>{}, k0 + BC, brs);
          }
        }
      });
      if constexpr (MK == 1) {
        // (short PV loops — small head dims — do not have a slot for every bias piece: the rest go out here)
        constexpr int kBStep = kStep2 >= 2 ? kStep2 : 2;
        static_for<kBtMax>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          if constexpr (i * kBStep + kBStep / 2 >= N2) {
            if (i < b_next) issue_bias(ic, k0 + BC, brs);
          }
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    FFPA_TSTAMP(4);  // PV loop
    // barrier B: every wave is done reading V(j); K(j+1) has landed and is visible
    if (pf_on) {  // (wave-uniform)
      issue_prefetch(k0 + FFPA_M16_PF_DIST * BC);
      dma_wait_except<1>();
    } else {
      dma_wait_except<0>();
    }
    __syncthreads();
    FFPA_TSTAMP(5);  // K(j+1) drain + wait at barrier B
  }
  if constexpr (kRowShare) {
    // the epilogue wants both row halves' running max and this lane's share of both row sums: the other half's come from its owner (the slots are free:
    // their last readers passed barrier B of the last step)
    *(FFPA_LDS float*)(xw_nd2 + dh * 1024) = m_run[0];
    *(FFPA_LDS float*)(xw_nd2 + (2 + dh) * 1024) = l_run[0];
    __syncthreads();
    m_run[0] = *(FFPA_LDS const float*)(xb_nd2);
    m_run[1] = *(FFPA_LDS const float*)(xb_nd2 + 4096 + 1024);
    l_run[0] = *(FFPA_LDS const float*)(xb_nd2 + 2048);
    l_run[1] = *(FFPA_LDS const float*)(xb_nd2 + 4096 + 3072);
  }
  }  // (the loop of the builds without the softmax pipeline)

  if (pf_on || kPipe) {  // the last touches land before their destination register is given to anything else (pipelined loop: and the
    dma_wait_all();      // zero-filled K pieces of the tile past the last one before the workgroup's LDS is)
    asm volatile("" : : "v"(pf_junk));
  }
  // ================= epilogue (prefill.cuh:1018-1093) =================
  asm volatile("s_nop 15\n\ts_nop 3");  // last PV MFMA (inline asm) -> accumulator reads below: wait states the compiler cannot see
  float l_tot[2], inv[2];
  l_tot[0] = l_run[0];
  l_tot[1] = l_run[1];
  row4_reduce2<false>(l_tot[0], l_tot[1]);
#pragma unroll
  for (int rh = 0; rh < 2; ++rh) inv[rh] = __builtin_amdgcn_rcpf(l_tot[rh]);  // fully masked row: 0 * inf = NaN, as SDPA
  if (a.nsplit > 1) {
    // split-KV partial: normalised fp32 O and its LSE (merged by ffpa_fwd_merge_kernel)
#pragma unroll
    for (int rh = 0; rh < 2; ++rh) {
      if (qrow[rh] >= a.Nq) continue;
      const bool dead = !(l_tot[rh] > 0.f);
      const int64_t prow = (((int64_t)split * a.B + b) * a.Hq + hq) * a.Nq + qrow[rh];
      float* wp = a.ws_o + prow * D + dh * DW + 4 * c;
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
        f32x4 w;
#pragma unroll
        for (int t = 0; t < 4; ++t) w[t] = dead ? 0.f : oacc[db][rh][t] * inv[rh];
        *(f32x4*)(wp + db * 16) = w;
      }
      // (one explicit FMA: left to -ffp-contract, builds of this kernel differed in whether they fused it — 1 ulp of the partial's LSE, which the merge
      // turns into an output ulp here and there; every build must produce the same bits for the same scores)
      if (c == 0 && dh == 0) a.ws_lse[prow] = dead ? -INFINITY : __builtin_fmaf(m_run[rh], 0.6931471805599453f, __logf(l_tot[rh]));
    }
    continue;
  }
  {
    // lanes c (even) and c + 1 trade one 4-column group per block: the even lane ends up with columns 16 db + 4 c .. + 8 of row
    // n, the odd one with the same columns of row 16 + n (v_permlane16_swap: odd 16-lane rows of the first operand <-> even rows
    // of the second)
    const int rsel = c & 1;
    const int orow = rsel ? qrow[1] : qrow[0];
    T* op = (T*)a.o + b * a.so[0] + hq * a.so[1] + (int64_t)(orow < a.Nq ? orow : 0) * a.so[2] + dh * DW + 4 * (c & ~1);
    const bool ok = orow < a.Nq;
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
      v4 g0, g1;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        g0[t] = (T)(oacc[db][0][t] * inv[0]);
        g1[t] = (T)(oacc[db][1][t] * inv[1]);
      }
      const u32x2 x0 = __builtin_bit_cast(u32x2, g0), x1 = __builtin_bit_cast(u32x2, g1);
      u32x4 run;
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const auto sw = __builtin_amdgcn_permlane16_swap(x0[w], x1[w], false, false);
        run[w] = sw[0];
        run[2 + w] = sw[1];
      }
      if (ok && dh * DW + db * 16 + 4 * (c & ~1) < a.d_valid) *(u32x4*)(op + db * 16) = run;
    }
#ifdef FFPA_M16_TIMING
    if (a.lse != nullptr && lane == 0) {  // 16 floats per wave at LSE row q0 + 16 * wave: six phase totals, whole kernel, KV tiles, two more phases
      float* tp = a.lse + ((int64_t)b * a.Hq + hq) * a.Nq + q0 + 16 * wave;
      for (int i = 0; i < 6; ++i) tp[i] = (float)tacc[i];
      tp[6] = (float)(__builtin_amdgcn_s_memtime() - tstart);
      tp[7] = (float)(nt - t0);
      tp[8] = (float)tacc[6];
      tp[9] = (float)tacc[7];
    }
#else
    if (a.lse != nullptr && c == 0 && dh == 0) {
#pragma unroll
      for (int rh = 0; rh < 2; ++rh)
        if (qrow[rh] < a.Nq) a.lse[((int64_t)b * a.Hq + hq) * a.Nq + qrow[rh]] = __builtin_fmaf(m_run[rh], 0.6931471805599453f, __logf(l_tot[rh]));
    }
#endif
  }
  }  // (pass: the second row tile of a paired workgroup)
}

}  // namespace ffpa
