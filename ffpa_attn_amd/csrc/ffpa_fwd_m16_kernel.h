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

#include <type_traits>

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

// The same wrappers with the NON-TEMPORAL hint on the riding piece (the packed-sequence kernel's decode-batch build: K / V bytes that ONE workgroup reads once).
template <typename T>
struct Mfma16Nt;
#define FFPA_MFMA16_NT(TYPE, SUFFIX)                                                                                                                        \
  template <>                                                                                                                                              \
  struct Mfma16Nt<TYPE> : Mfma16<TYPE> {                                                                                                                   \
    typedef Elem<TYPE>::v8 v8;                                                                                                                             \
    template <int KIND, int LCONST>                                                                                                                        \
    static __device__ __forceinline__ void with_dma(f32x4& d, v8 a, v8 b, u32x4 rsrc, uint32_t lds_base, uint32_t voff, uint32_t soff) {                  \
      if constexpr (KIND == 0)                                                                                                                             \
        asm volatile("s_add_u32 m0, %3, %7\n\tv_mfma_f32_16x16x32_" SUFFIX " %0, %1, %2, 0\n\tbuffer_load_dwordx4 %4, %5, %6 offen nt lds"                \
                     : "=&v"(d) : "v"(a), "v"(b), "s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff), "n"(LCONST) : "memory", "scc" FFPA_M0_CLOBBER);          \
      else if constexpr (KIND == 1)                                                                                                                        \
        asm volatile("s_add_u32 m0, %3, %7\n\tv_mfma_f32_16x16x32_" SUFFIX " %0, %1, %2, %0\n\tbuffer_load_dwordx4 %4, %5, %6 offen nt lds"               \
                     : "+v"(d) : "v"(a), "v"(b), "s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff), "n"(LCONST) : "memory", "scc" FFPA_M0_CLOBBER);           \
      else                                                                                                                                                 \
        asm volatile("s_add_u32 m0, %3, %7\n\tv_mfma_f32_16x16x32_" SUFFIX " %0, %1, %2, %0\n\tbuffer_load_dwordx4 %4, %5, %6 offen nt lds"               \
                     : "+a"(d) : "v"(a), "v"(b), "s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff), "n"(LCONST) : "memory", "scc" FFPA_M0_CLOBBER);           \
    }                                                                                                                                                      \
  };
FFPA_MFMA16_NT(__bf16, "bf16")
FFPA_MFMA16_NT(_Float16, "f16")
#undef FFPA_MFMA16_NT

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
#define FFPA_M16_MFMA Mfma16<T>
#define FFPA_M16_DMA16 lds_dma_16_at
#include "ffpa_fwd_m16_head.inc"
  // workgroup -> (batch, head, row tile, split): as ffpa_fwd_split_d_kernel (all row tiles of a head on one XCD)
  int vid = blockIdx.x;
  if (!(a.flags & kFlagNoXcdRemap)) vid = xcd_logical_id(vid, gridDim.x, a.xcd_group);
  const int split = vid % a.nsplit;
  vid /= a.nsplit;
  const int bh = vid / a.nqt;
  int qt = vid - bh * a.nqt;
  // longest rows first: with the causal flag — and with mask ranges, whose usual source is a causal-like boolean mask (later rows see
  // more keys; for any other mask the order of a head's row tiles does not matter) — so that the launch ends on its short workgroups
  if (a.causal || (MASK && a.kv_bounds != nullptr)) qt = a.nqt - 1 - qt;
#define FFPA_M16_TILE_DONE return
#define FFPA_M16_ROW_INV(l) __builtin_amdgcn_rcpf(l)
#define FFPA_M16_ROW_OUT(x, rh) (T)((x) * inv[rh])
#define FFPA_M16_LSE_INDEX(row) ((int64_t)b * a.Hq + hq) * a.Nq + (row)
#define FFPA_M16_WS_ROW(row) ((((int64_t)split * a.B + b) * a.Hq + hq) * a.Nq + (row))
#define FFPA_M16_Q_ROW_OFF(row) ((int64_t)(row) * a.sq[2])
#define FFPA_M16_O_ROW_OFF(row) ((int64_t)(row) * a.so[2])
#include "ffpa_fwd_m16_tile.inc"
#undef FFPA_M16_O_ROW_OFF
#undef FFPA_M16_Q_ROW_OFF
#undef FFPA_M16_WS_ROW
#undef FFPA_M16_LSE_INDEX
#undef FFPA_M16_ROW_OUT
#undef FFPA_M16_ROW_INV
#undef FFPA_M16_TILE_DONE
#undef FFPA_M16_DMA16
#undef FFPA_M16_MFMA
}

// PAIRED ROW TILES (round 6: launches under the causal flag, FwdArgs::pair_tiles): workgroup i of a head walks row tile nqt - 1 - i (its long one) and then
// row tile i (the short one) — every workgroup of the launch walks nqt + 1 diagonal-bounded tiles' worth of KV steps, the launch has half the workgroups
// (half the dispatches, no tail of short workgroups), and per row nothing changes: the same tile, the same recurrence, the same bits.  The whole tile —
// tile range, Q fragments, KV loop, epilogue — runs once per pass; LDS is free between the passes (every wave has passed the last step's barrier B, and the
// pipelined loop's own post-loop barrier, before any wave can start the next pass's prologue DMA).  A kernel of its own, from the same text, so that the
// one-tile kernels stay exactly what they were (a loop around their body costs the register allocator: - 0.2 ... - 1.2 % on launches that do not pair).
template <typename T, int D, bool DROP = false>
__global__ __launch_bounds__(256) void ffpa_fwd_m16_pair_kernel(const FwdArgs a) {
  constexpr int MK = 0;  // the builds without a bias: what a launch under the causal flag runs
#define FFPA_M16_MFMA Mfma16<T>
#define FFPA_M16_DMA16 lds_dma_16_at
#include "ffpa_fwd_m16_head.inc"
  int vid = blockIdx.x;
  if (!(a.flags & kFlagNoXcdRemap)) vid = xcd_logical_id(vid, gridDim.x, a.xcd_group);
  const int split = vid % a.nsplit;  // (KV-split launches never pair: nsplit == 1)
  vid /= a.nsplit;
  const int nqt_wg = a.pair_tiles ? (a.nqt + 1) >> 1 : a.nqt;
  const int bh = vid / nqt_wg;
  const int qt_wg = vid - bh * nqt_wg;
  const int npass = (a.pair_tiles && 2 * qt_wg != a.nqt - 1) ? 2 : 1;  // (an odd tile count: the middle tile is its own partner)
  for (int pass = 0; pass < npass; ++pass) {
    const int qt = pass == 0 ? a.nqt - 1 - qt_wg : qt_wg;
#define FFPA_M16_TILE_DONE continue
#define FFPA_M16_ROW_INV(l) __builtin_amdgcn_rcpf(l)
#define FFPA_M16_ROW_OUT(x, rh) (T)((x) * inv[rh])
#define FFPA_M16_LSE_INDEX(row) ((int64_t)b * a.Hq + hq) * a.Nq + (row)
#define FFPA_M16_WS_ROW(row) ((((int64_t)split * a.B + b) * a.Hq + hq) * a.Nq + (row))
#define FFPA_M16_Q_ROW_OFF(row) ((int64_t)(row) * a.sq[2])
#define FFPA_M16_O_ROW_OFF(row) ((int64_t)(row) * a.so[2])
#include "ffpa_fwd_m16_tile.inc"
#undef FFPA_M16_O_ROW_OFF
#undef FFPA_M16_Q_ROW_OFF
#undef FFPA_M16_WS_ROW
#undef FFPA_M16_LSE_INDEX
#undef FFPA_M16_ROW_OUT
#undef FFPA_M16_ROW_INV
#undef FFPA_M16_TILE_DONE
#undef FFPA_M16_DMA16
#undef FFPA_M16_MFMA
  }
}

// PACKED SEQUENCES (round 6, the reference's ffpa_attn_varlen_func: src/ffpa_attn/ffpa_attn_interface.py:192-279, served there by its CuTe-DSL backend only —
// src/ffpa_attn/cute/__init__.py:466-575): q [T_q, Hq, D], k / v [T_k, Hkv, D], o [T_q, Hq, D], LSE [Hq, T_q], sequence i owning rows cu_q[i] .. cu_q[i + 1] of q / o and
// cu_k[i] .. cu_k[i + 1] of k / v.  ONE launch for the whole batch, the boundaries read on the DEVICE (two scalar loads per array and workgroup: no host round trip,
// so the call captures into a HIP graph): the grid holds ceil(max_seqlen_q / BR) row tiles per (sequence, head); a workgroup looks its sequence up, leaves when its row
// tile lies past the sequence's last row, and otherwise runs the SAME tile text as the dense kernel on a FwdArgs whose base pointers, lengths and causal offset
// (tail-aligned per sequence: Nkv_i - Nq_i) are that sequence's — per row the same recurrence, the same bits as a dense launch of that sequence alone (the launch
// plan's other tiles — wide rows, KV splits, pairs — are not used here).  Rows no key of which is visible (an empty key range; under the causal flag the first
// Nq_i - Nkv_i rows of a sequence with more queries than keys) come out as O = 0, LSE = -inf: the reference's contract for this entry point
// (tests/test_ffpa_cute_sm100.py:1117-1183), where the dense kernel keeps SDPA's NaN.
struct VarlenArgs {
  const int* cu_q;       // [batch + 1] row offsets into q / o.  NULL = a DENSE launch that only borrows this kernel's workgroup order (see the kernel)
  const int* cu_k;       // [batch + 1] row offsets into k / v
  int64_t lse_stride_h;  // elements between two heads of the LSE tensor (>= T_q)
  int head_chunk;        // consecutive query heads that walk a sequence side by side (the workgroup order below): a divisor of Hq
  // SHORT query sequences under GQA (decode: one token per sequence; speculative decoding / multi-token prediction / small prefill chunks: a few): pack > 0 = the
  // `pack` query heads of a KV group x the sequence's tokens are the ROWS of ONE tile, head-major — row r of sequence i is (head r / ntok_i, token r % ntok_i), the
  // layout the dense path's host-side packing produces (FwdArgs::causal_row_mod = ntok_i: the causal limit of row r is r % ntok_i + causal_offset).  FwdArgs then
  // describes Hkv "heads" (head stride = one KV group, row stride = one query head): the group's K / V stream is read by ONE workgroup instead of `pack` (the
  // reference's pack_gqa, src/ffpa_attn/cute/__init__.py:792-829; the dense path packs the same way for Nq <= 7: hip/__init__.py).  A token's address is its row
  // offset times the TOKEN stride below, not the (head) row stride.  The launch side packs when pack x max_seqlen_q rows fit one tile.  0 = rows are tokens.
  int pack;
  int64_t q_tok_stride, o_tok_stride;  // elements between two tokens of q / o
  const int* used_k;     // optional [batch]: sequence i uses only the first used_k[i] of its key rows (a KV cache of fixed capacity per sequence whose valid
                         // length lives on the device: FlashAttention's seqused_k / cache_seqlens); NULL = all of cu_k[i] .. cu_k[i + 1]
  // KV SPLITS (FwdArgs::nsplit > 1: launches of one row tile per (sequence, head), and prefill launches that leave most of the chip idle): workgroup (pair, row tile,
  // split) walks KV tiles [split * tps, (split + 1) * tps) of ITS sequence, tps = ceil(tiles that row tile of that sequence can see / nsplit) computed here on the device, and stores a normalised fp32 partial + LSE to FwdArgs::ws_o / ws_lse at row
  // split * ws_split_rows + head * ws_head_rows + token (ffpa_varlen_merge_kernel, ffpa_varlen_merge.h, combines them: the reference's decode stage 2)
  int64_t ws_head_rows;   // total_q
  int64_t ws_split_rows;  // query heads x total_q
  // COMPACT grid (packed launches of several row tiles per head whose caller says total_q): > 0 = the grid holds this many row-tile slots per head — an upper
  // bound of sum_i ceil(len_i / block rows), ceil(total_q / block rows) + batch — instead of batch x ceil(max_seqlen_q / block rows): a ragged batch sizes its grid
  // by the rows there are, not by its longest sequence (8 sequences of 256 ... 4864 tokens: 304 slots per head of which 128 hold rows -> 136).  Slot -> (sequence,
  // row tile) on the device: the waves scan the sequences' tile counts 64 at a time (one load per lane, a wave prefix sum, a ballot); same order as the full grid.
  int compact_tiles;
};

// NT: the decode-batch build — every K / V piece carries the non-temporal hint (each byte has ONE reader and the batch's K + V do not fit the caches: the launch
// side's rule, ffpa_capi.hip; LDS-DMA from HBM 5.9 -> 7.3 TB/s with it, profiles/r04_kv_stream.txt) — otherwise the same kernel.
template <typename T, int D, bool NT = false>
__global__ __launch_bounds__(256) void ffpa_fwd_m16_varlen_kernel(const FwdArgs a_in, const VarlenArgs va) {
  constexpr int MK = 0;  // no attn_bias, no mask ranges: what the reference's packed entry point accepts
  constexpr bool DROP = false;
#define FFPA_M16_MFMA std::conditional_t<NT, Mfma16Nt<T>, Mfma16<T>>
#define FFPA_M16_DMA16 LdsDma16<NT>::template at
#include "ffpa_fwd_m16_head.inc"
  int vid = blockIdx.x;
  if (!(a_in.flags & kFlagNoXcdRemap)) vid = xcd_logical_id(vid, gridDim.x, a_in.xcd_group);
  int split = 0;
  if (a_in.nsplit > 1) {  // (the KV ranges of a pair are neighbours in the launch order: one XCD, one after the other)
    const int pair = vid / a_in.nsplit;
    split = vid - pair * a_in.nsplit;
    vid = pair;
  }
  // Order of the (sequence, head) pairs: head CHUNK-major, then sequence, then the head inside its chunk (va.head_chunk consecutive heads: Hq / 8 when that is
  // whole, else 1).  The XCD remap hands every XCD a contiguous range of pairs (all row tiles of a pair on one XCD: its K / V stream stays in one L2), and
  // sequences differ in length by orders of magnitude — sequence-major (the dense order) gives one XCD the longest sequence and another the shortest (measured:
  // 400 vs 960 TFLOPS on the bench's 256 ... 4864-token batch).  Chunk-major gives every XCD the same heads of EVERY sequence; and the heads of a chunk — under
  // GQA heads of ONE KV group — walk the same sequence side by side, so that the group's K / V stream is fetched once per L2, not once per head
  // ... side by side at the level of ROW TILES: (chunk, sequence, row tile, head in chunk) — a head's tiles alone fill an XCD's 32 CUs for a whole round, so heads
  // that merely follow each other stream the sequence's K / V once each (measured: no fewer HBM bytes than head-major order); interleaved per tile, the same row
  // tile of the chunk's heads runs at the same time on the same keys
  int chunk, seq, qt, head_in_chunk;
  int seq_tiles = a_in.nqt;  // row tiles of this sequence in the grid (compact grid: the sequence's own count)
  if (va.compact_tiles > 0) {
    const int per_chunk = va.compact_tiles * va.head_chunk;
    chunk = vid / per_chunk;
    const int in_chunk = vid - chunk * per_chunk;
    const int slot = in_chunk / va.head_chunk;
    head_in_chunk = in_chunk - slot * va.head_chunk;
    // slot -> (sequence, row tile): the sequences' tile counts, 64 sequences per step
    seq = -1, qt = 0;
    int before = 0;
    for (int s0 = 0; s0 < a_in.B; s0 += 64) {
      const int i = s0 + lane;
      int n = 0;
      if (i < a_in.B) {
        const int len = va.cu_q[i + 1] - va.cu_q[i];
        n = len > 0 ? (len + BR - 1) / BR : 0;
      }
      int incl = n;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
      }
      const int total = __shfl(incl, 63);
      if (slot < before + total) {
        const unsigned long long m = __ballot(slot < before + incl);
        const int l = __ffsll((long long)m) - 1;
        seq = s0 + l;
        seq_tiles = __shfl(n, l);
        qt = slot - (before + __shfl(incl, l) - seq_tiles);
        break;
      }
      before += total;
    }
    if (seq < 0) return;  // (a slot past the batch's last row tile: the bound is not tight)
    seq = __builtin_amdgcn_readfirstlane(seq);
    seq_tiles = __builtin_amdgcn_readfirstlane(seq_tiles);
    qt = __builtin_amdgcn_readfirstlane(qt);
  } else {
    const int per_seq = a_in.nqt * va.head_chunk, per_chunk = a_in.B * per_seq;
    chunk = vid / per_chunk;
    const int in_chunk = vid - chunk * per_chunk;
    seq = in_chunk / per_seq;
    const int in_seq = in_chunk - seq * per_seq;
    qt = in_seq / va.head_chunk;
    head_in_chunk = in_seq - qt * va.head_chunk;
  }
  if (a_in.causal) qt = seq_tiles - 1 - qt;  // longest rows first
  const int bh = seq * a_in.Hq + chunk * va.head_chunk + head_in_chunk;
  FwdArgs a = a_in;
  int q_lo;  // packed: the sequence's first row of q / o (LSE [Hq, T_q]: its column); dense: the batch element's first LSE row
  int ntok = 1;  // tokens of this sequence (>= 1): packed rows are (row / ntok, row % ntok) = (head of the group, token)
  if (va.cu_q == nullptr) {
    // DENSE launches in this kernel's workgroup order (ffpa_attn_fwd -> ffpa_capi.hip: causal + GQA, no bias, no dropout, every row sees a key): the
    // arguments are the dense call's as they are — "sequence" = batch element, batch strides live —, only the order of the workgroups is this kernel's
    q_lo = seq * a_in.Hq * a_in.Nq;
    // (KV ranges in this mode: a causal launch of one round of workgroups whose long row tiles would run alone at the end — ffpa_capi.hip pick_tile_ranges; the
    // workspace rows are the dense call's [split, batch, head, row]: ws_head_rows = Nq, ws_split_rows = B x Hq x Nq, merged by ffpa_fwd_merge_kernel)
  } else {
    q_lo = va.cu_q[seq];
    const int k_lo = va.cu_k[seq];
    const int ntok_seq = va.cu_q[seq + 1] - q_lo;
    int nkv_seq = va.cu_k[seq + 1] - k_lo;
    if (va.used_k != nullptr) {
      const int used = va.used_k[seq];
      nkv_seq = nkv_seq < used ? nkv_seq : used;
    }
    ntok = ntok_seq > 0 ? ntok_seq : 1;
    const int nq_seq = va.pack ? va.pack * ntok_seq : ntok_seq;  // (packed: the rows of a sequence are (head of the group, token), head-major)
    if (qt * BR >= nq_seq) return;  // (max_seqlen_q sized the grid: this sequence is shorter)
    // (batch strides are zero: the launch side)
    a.Nq = nq_seq;
    a.Nkv = nkv_seq > 0 ? nkv_seq : 0;
    a.causal_offset = a.Nkv - ntok_seq;  // (tail-aligned per sequence; a single packed token runs without the causal flag — it sees every key of its sequence)
    if (va.pack) a.causal_row_mod = ntok_seq;
    if (a_in.nsplit > 1 && (int64_t)q_lo + ntok_seq > va.ws_head_rows) return;  // (a caller whose total_q is smaller than its boundaries say: nothing is stored outside the scratch it sized)
    a.q = (const T*)a_in.q + (int64_t)q_lo * va.q_tok_stride;
    a.o = (T*)a_in.o + (int64_t)q_lo * va.o_tok_stride;
    a.k = (const T*)a_in.k + (int64_t)k_lo * a_in.sk[2];
    a.v = (const T*)a_in.v + (int64_t)k_lo * a_in.sv[2];
  }
  if (a_in.nsplit > 1) {
    int tiles = (a.Nkv + BC - 1) / BC;
    if (a.causal) {
      // under the causal flag a row tile walks the KV tiles up to ITS diagonal (the tile text's clamp, restated): those are what its ranges share out —
      // every row tile of an under-filled prefill launch splits its own visible keys evenly (one-row-tile launches: all keys of the sequence, as before)
      const int last_row = a.causal_row_mod ? a.causal_row_mod - 1 : qt * BR + BR - 1;
      const int64_t last = (int64_t)last_row + a.causal_offset;
      const int ntc = last < 0 ? 0 : (int)(last / BC) + 1;
      tiles = tiles < ntc ? tiles : ntc;
    }
    a.tiles_per_split = (tiles + a_in.nsplit - 1) / a_in.nsplit;  // (fewer tiles than ranges leaves ranges empty: dead partials, weight 0 in the merge)
  }
#define FFPA_M16_TILE_DONE return
#define FFPA_M16_ROW_INV(l) ((l) > 0.f ? __builtin_amdgcn_rcpf(l) : 0.f)
#define FFPA_M16_ROW_OUT(x, rh) (l_tot[rh] > 0.f ? (T)((x) * inv[rh]) : (T)0.f)  // (the select BEHIND product + conversion: those stay the dense kernel's one instruction — fp16: v_fma_mixlo, one rounding — and its bits)
#define FFPA_M16_LSE_INDEX(row) (va.pack ? (int64_t)(hq * va.pack + (row) / ntok) * va.lse_stride_h + q_lo + (row) % ntok : (int64_t)hq * va.lse_stride_h + q_lo + (row))
// (the KV-split workspace of the packed call: [split, query head, token] rows — ffpa_varlen_merge_kernel reads them back by (head, token))
#define FFPA_M16_WS_ROW(row) ((int64_t)split * va.ws_split_rows + (va.pack ? (int64_t)(hq * va.pack + (row) / ntok) * va.ws_head_rows + q_lo + (row) % ntok : (int64_t)hq * va.ws_head_rows + q_lo + (row)))
// (packed rows: a.sq[2] / a.so[2] are the HEAD strides of q / o, a token is q_tok_stride / o_tok_stride further; rows are tokens: ntok-independent)
#define FFPA_M16_Q_ROW_OFF(row) (va.pack ? (int64_t)((row) / ntok) * a.sq[2] + (int64_t)((row) % ntok) * va.q_tok_stride : (int64_t)(row) * a.sq[2])
#define FFPA_M16_O_ROW_OFF(row) (va.pack ? (int64_t)((row) / ntok) * a.so[2] + (int64_t)((row) % ntok) * va.o_tok_stride : (int64_t)(row) * a.so[2])
#include "ffpa_fwd_m16_tile.inc"
#undef FFPA_M16_O_ROW_OFF
#undef FFPA_M16_Q_ROW_OFF
#undef FFPA_M16_WS_ROW
#undef FFPA_M16_LSE_INDEX
#undef FFPA_M16_ROW_OUT
#undef FFPA_M16_ROW_INV
#undef FFPA_M16_TILE_DONE
#undef FFPA_M16_DMA16
#undef FFPA_M16_MFMA
}

}  // namespace ffpa
