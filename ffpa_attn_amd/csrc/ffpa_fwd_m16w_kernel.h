// ffpa_fwd_m16w_kernel.h — the WIDE-ROW prefill tile of the small split-D head dims (D <= 320) on the 16x16x32 MFMA shape.
//
// Same algorithm and per-row recurrence as ffpa_fwd_m16_kernel (ffpa_fwd_m16_kernel.h, which follows the reference's split_d_fwd_sm80:
// csrc/cuffpa/native/sm_80/split_d.cuh:96-777; tile traits per head dim: csrc/cuffpa/cute/attn_traits.cuh:170-212, native/launch.cuh:69-104).
// What differs is how the tile is sized: to the REGISTER FILE instead of to D = 512.
//
//   * ffpa_fwd_m16_kernel gives every wave 32 query rows (two 16-row halves) at every head dim <= 512.  At D = 320 that is O^T = 160 of the
//     256 AGPRs, and every K / V^T fragment read from LDS — and every L2 -> LDS DMA byte — feeds two MFMAs.  Here a wave owns RH 16-row
//     halves, RH chosen so that O^T fills the accumulator file: D = 320 -> RH = 3 (240 AGPRs; Q fragments 120 VGPRs, S^T 48): a fragment
//     feeds THREE MFMAs, i.e. - 33 % LDS fragment reads, DMA pieces and DMA issue slots per FLOP.  This chip runs the kernel against a power
//     cap (profiles/NOTES.md section 3): bytes moved per FLOP are what the clock pays for.
//   * 64-key tiles (S^T of 128 keys x 48 rows would be 96 registers): the two tile images are 80 KiB at D = 320 — half of the LDS.  The
//     other half DOUBLE-BUFFERS them: tile j + 1 streams into the second pair of images while tile j is contracted, all of its pieces a
//     whole step ahead of their first reader.  The three workgroup barriers per KV step of the single-buffered pipeline (A1: K rows free,
//     A2: V landed, B: V rows free + K landed) collapse into ONE — at the top of a step: "tile j has landed everywhere, everyone is done
//     with tile j - 1" — and the one counted DMA wait of a step has the softmax and the whole PV loop behind the last piece it waits for.
//
// Builds: MK = 0 (no mask) and MK = 2 (boolean masks and / or mask ranges — BASELINE config 4 as specified); additive biases and dropout
// keep the 32-row tile, whose LDS has room for the bias caches and whose register file has room for the Philox state.  The launch side
// (ffpa_capi.hip) takes this tile when a launch's rounds of workgroups come out cheaper with it (192-row tiles quantise a launch differently
// from 128-row ones); FFPA_FLAG_WIDE_TILE / FFPA_FLAG_NO_WIDE_TILE force the choice for A/B runs and tests.
#pragma once

#include "ffpa_fwd_m16_kernel.h"

#ifndef FFPA_M16W_PF1
#define FFPA_M16W_PF1 3  // K fragments requested ahead of their (RH) MFMAs (3 / 3 vs 4 / 4 vs 6 / 4: + 1 % / 0 / - 0.5 % on config 4, profiles/r05_wide_tile.txt)
#endif
#ifndef FFPA_M16W_PF2
#define FFPA_M16W_PF2 3  // V^T fragments (two transpose reads each) requested ahead
#endif
#ifndef FFPA_M16W_QK_PIECES
#define FFPA_M16W_QK_PIECES 16  // sixteenths of a step's 2 PPW DMA pieces that ride on the QK^T loop's fragments (the rest: on the PV loop's, front-loaded)
#endif

namespace ffpa {

#ifndef FFPA_M16W_DIMS
#define FFPA_M16W_DIMS(D) ((D) == 320)  // head dims whose library object carries the wide-row tile
#endif
constexpr bool m16w_available(int D) { return FFPA_M16W_DIMS(D); }
// rows per workgroup / LDS bytes of ffpa_fwd_m16w_kernel<., D, RH, .> (the launch plan uses the same rules)
constexpr int m16w_block_rows(int RH) { return 64 * RH; }
// keys per tile: 64 where four images fit the LDS (D <= 320), else 32
constexpr int m16w_block_keys(int D) { return 4 * 64 * D * 2 <= 160 * 1024 ? 64 : 32; }
constexpr int m16w_lds_bytes(int D) { return 4 * m16w_block_keys(D) * D * 2; }
// 16-row halves per wave: as many as the 256 AGPRs hold of O^T (D / 16 blocks x 4 registers per half), at most 4 (Q fragments: D / 8 VGPRs per half)
#ifndef FFPA_M16W_MAX_RH
#define FFPA_M16W_MAX_RH 4
#endif
constexpr int m16w_row_halves(int D) { return 256 / (D / 4) > FFPA_M16W_MAX_RH ? FFPA_M16W_MAX_RH : 256 / (D / 4); }
#ifdef FFPA_PRODUCT_BUILD  // (as in ffpa_fwd_m16_kernel.h: the product library is built from the shipped values only)
#if FFPA_M16W_PF1 != 3 || FFPA_M16W_PF2 != 3 || FFPA_M16W_QK_PIECES != 16 || FFPA_M16W_MAX_RH != 4
#error "FFPA_PRODUCT_BUILD: a developer switch of the wide-row tile is not at its shipped default"
#endif
#endif

template <bool IS_MAX, int RH>
__device__ __forceinline__ void row4_reduce_n(float (&t)[RH]) {
#pragma unroll
  for (int i = 0; i + 1 < RH; i += 2) row4_reduce2<IS_MAX>(t[i], t[i + 1]);
  if constexpr (RH % 2 == 1) row4_reduce1<IS_MAX>(t[RH - 1]);
}

template <typename T, int D, int RH, int BC, int MK = 0>
__global__ __launch_bounds__(256) void ffpa_fwd_m16w_kernel(const FwdArgs a) {
  static_assert(MK == 0 || MK == 2, "mask kinds: 0 = none, 2 = boolean mask / mask ranges");
  constexpr bool MASK = MK == 2;
  using E = Elem<T>;
  using M = Mfma16<T>;
  using v8 = typename E::v8;
  using v4 = typename E::v4;
  static_assert(D % 64 == 0 && RH >= 2 && RH <= 4 && (D / 16) * RH * 4 <= 256, "O^T (D / 4 registers per 16-row half) must fit the AGPRs");
  static_assert(BC == 32 || BC == 64, "32- or 64-key tiles");
  constexpr int WR = 16 * RH, BR = 4 * WR;
  constexpr int KS = D / 32;    // QK contraction steps
  constexpr int NKB = BC / 16;  // 16-key S^T blocks per tile
  constexpr int NKS = BC / 32;  // PV contraction steps per tile
  constexpr int NDB = D / 16;   // 16-column O^T blocks
  constexpr int RB = D * 2;
  constexpr int TILE = BC * RB;
  constexpr int PPW = TILE / 4096;  // 1 KiB DMA pieces per wave per tile image
  static_assert(TILE % 4096 == 0 && 4 * TILE <= 160 * 1024, "two double-buffered tile images in the LDS");
  constexpr int N1 = KS * NKB;   // K fragments per tile
  constexpr int N2 = NDB * NKS;  // V^T fragments per tile
  constexpr int PF1 = FFPA_M16W_PF1, PF2 = FFPA_M16W_PF2;
  constexpr int NP = 2 * PPW;                                // DMA pieces per wave and step
  constexpr int cntQ = NP * FFPA_M16W_QK_PIECES / 16, cntP = NP - cntQ;
  static_assert(cntQ >= 0 && cntP >= 0 && cntQ <= N1 && cntP <= N2, "at most one piece per fragment");
  constexpr int stepP = cntP > 0 ? (N2 / 2 / cntP > 0 ? N2 / 2 / cntP : 1) : 1;  // PV pieces sit in the loop's first half (they are awaited at its end)
  static_assert(cntP == 0 || (cntP - 1) * stepP < N2, "every piece of the PV loop must ride on one of its fragments");
  constexpr int KV = (D % 128 == 0) ? 4 : 2;        // K fragment address variants (the swizzle's reach, see ffpa_fwd_m16_kernel.h)
  constexpr int KVB = (D % 128 == 0) ? 256 : 128;
  constexpr int VV = (D % 128 == 0) ? 8 : 4;        // V^T fragment address variants
  constexpr int VVB = (D % 128 == 0) ? 256 : 128;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  FFPA_LDS char* const Kt = (FFPA_LDS char*)smem;  // K images: Kt, Kt + TILE
  FFPA_LDS char* const Vt = Kt + 2 * TILE;         // V images: Vt, Vt + TILE

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15;
  const int c = lane >> 4;

  // workgroup -> (batch, head, row tile, split): as ffpa_fwd_m16_kernel (all row tiles of a head on one XCD, longest rows first)
  int vid = blockIdx.x;
  if (!(a.flags & kFlagNoXcdRemap)) vid = xcd_logical_id(vid, gridDim.x, a.xcd_group);
  const int split = vid % a.nsplit;
  vid /= a.nsplit;
  const int bh = vid / a.nqt;
  int qt = vid - bh * a.nqt;
  if (a.causal || (MASK && a.kv_bounds != nullptr)) qt = a.nqt - 1 - qt;
  const int b = bh / a.Hq;
  const int hq = bh - b * a.Hq;
  const int hkv = hq / a.group;
  const int q0 = qt * BR;
  const int wq0 = q0 + wave * WR;
  int qrow[RH], qrow_c[RH];
#pragma unroll
  for (int rh = 0; rh < RH; ++rh) {
    qrow[rh] = wq0 + 16 * rh + n16;
    qrow_c[rh] = qrow[rh] < a.Nq ? qrow[rh] : a.Nq - 1;
  }

  const T* __restrict__ Kg = (const T*)a.k + b * a.sk[0] + hkv * a.sk[1];
  const T* __restrict__ Vg = (const T*)a.v + b * a.sv[0] + hkv * a.sv[1];
  const uint32_t k_row_bytes = (uint32_t)a.sk[2] * 2u;
  const uint32_t v_row_bytes = (uint32_t)a.sv[2] * 2u;

  // ---- LDS-DMA: piece p = wave * PPW + i covers slots [64 p, 64 p + 64) of the row-major image (16-byte slot s of row `key` stored at slot
  // s ^ swizzle(key): applied on the per-lane SOURCE offset, the destination is lane-linear); the per-lane source offsets are tile-invariant.
  // A caller's head dim below D: columns at and past it read as zeros (out-of-range offset -> the descriptor's range check zero-fills).
  const uint32_t rb_valid = (uint32_t)a.d_valid * 2u;
  const int slots_valid = a.d_valid >> 3;
  uint32_t krel[PPW], vrel[PPW];
  {
    constexpr int SPR = D / 8;  // 16-byte slots per row
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int g = (wave * PPW + i) * 64 + lane;
      const int key = g / SPR;
      const int slot = g - key * SPR;
      const int vs = slot ^ m16_v_swizzle<D>(key);
      vrel[i] = (uint32_t)key * v_row_bytes + (uint32_t)(vs << 4);
      if (vs >= slots_valid) vrel[i] = kDmaOob;
      const int ks = slot ^ m16_k_swizzle<D>(key);
      krel[i] = (uint32_t)key * k_row_bytes + (uint32_t)(ks << 4);
      if (ks >= slots_valid) krel[i] = kDmaOob;
    }
  }
  // this wave's pieces of image `buf` land at base + buf TILE + i KiB
  const uint32_t k_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(uintptr_t)Kt + (uint32_t)(wave * PPW * 1024)));
  const uint32_t v_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(uintptr_t)Vt + (uint32_t)(wave * PPW * 1024)));
  // piece t of a step (t < PPW: K, else V — alternating K / V pieces measured the same: word1 in profiles/r05_wide_tile.txt), alone or riding on an MFMA (kind: 0 first of a chain, 1 accumulate in
  // VGPRs, 2 accumulate in an AGPR tile)
  auto piece_is_k = [](int t) constexpr { return t < PPW; };
  auto piece_idx = [](int t) constexpr { return t < PPW ? t : t - PPW; };
  auto issue_piece = [&](auto tc, int key0, uint32_t buf_off) {
    constexpr int t = decltype(tc)::value;
    constexpr int i = piece_idx(t);
    if constexpr (piece_is_k(t)) {
      const TileSrc ts = tile_src<BC>(Kg, k_row_bytes, key0, a.Nkv, rb_valid);
      lds_dma_16_at<i * 1024>(ts.rsrc, k_lds + buf_off, krel[i], 0u);
    } else {
      const TileSrc ts = tile_src<BC>(Vg, v_row_bytes, key0, a.Nkv, rb_valid);
      lds_dma_16_at<i * 1024>(ts.rsrc, v_lds + buf_off, vrel[i], 0u);
    }
  };
  auto issue_piece_on = [&](auto tc, int key0, uint32_t buf_off, auto kindc, f32x4& d, v8 fa, v8 fb) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value;
    constexpr int i = piece_idx(t);
    if constexpr (piece_is_k(t)) {
      const TileSrc ts = tile_src<BC>(Kg, k_row_bytes, key0, a.Nkv, rb_valid);
      M::template with_dma<decltype(kindc)::value, i * 1024>(d, fa, fb, ts.rsrc, k_lds + buf_off, krel[i], 0u);
    } else {
      const TileSrc ts = tile_src<BC>(Vg, v_row_bytes, key0, a.Nkv, rb_valid);
      M::template with_dma<decltype(kindc)::value, i * 1024>(d, fa, fb, ts.rsrc, v_lds + buf_off, vrel[i], 0u);
    }
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
  // mask ranges (ffpa_fwd_params.kv_bounds, four ints per block of 32 query rows: [first, end) visible to some row, [free_lo, free_hi) visible
  // to EVERY row): KV tiles no row of this row tile can see are skipped; tiles inside the intersection of the free ranges of the 32-row blocks
  // this wave's rows touch do not read the mask at all
  int free_lo = 0, free_hi = 0;
  if (MASK && a.kv_bounds != nullptr) {
    const int* bp = a.kv_bounds + b * a.s_bounds[0] + hq * a.s_bounds[1];
    int first = 0x7fffffff, end = 0;
    static_assert(BR % 32 == 0, "row tiles are whole 32-row blocks of the mask ranges");
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
    t0 = __builtin_amdgcn_readfirstlane(t0 > tf ? t0 : tf);
    nt = __builtin_amdgcn_readfirstlane(nt < te ? nt : te);
    int flo = 0, fhi = 0x7fffffff;
    constexpr int kBlk = (WR + 31) / 32 + (WR % 32 != 0 ? 1 : 0);  // 32-row blocks a wave's rows can touch
#pragma unroll
    for (int blk = 0; blk < kBlk; ++blk) {
      const int r32 = wq0 / 32 + blk;
      if (r32 * 32 < wq0 + WR && r32 * 32 < a.Nq) {
        const int lo = bp[4 * r32 + 2], hi = bp[4 * r32 + 3];
        flo = flo > lo ? flo : lo;
        fhi = fhi < hi ? fhi : hi;
      }
    }
    free_lo = __builtin_amdgcn_readfirstlane(flo);
    free_hi = __builtin_amdgcn_readfirstlane(fhi);  // (a wave whose rows all lie past the last query row: everything is "free", nothing it computes is stored)
  }

  // ---- Q fragments (B operand of S^T): lane (n, c) holds Q[row 16 rh + n][32 s + 8 c .. + 8]
  v8 qf[KS][RH];
#pragma unroll
  for (int rh = 0; rh < RH; ++rh) {
    const T* qp = (const T*)a.q + b * a.sq[0] + hq * a.sq[1] + (int64_t)qrow_c[rh] * a.sq[2] + c * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      qf[s][rh] = (s * 32 + c * 8 < a.d_valid && a.q_mode != 1) ? *(const v8*)(qp + s * 32) : __builtin_bit_cast(v8, z);
    }
  }
  if (a.q_mode == 2) {  // a negative softmax scale reaches the kernel as (-Q, |scale|)
#pragma unroll
    for (int rh = 0; rh < RH; ++rh)
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        u32x4 w = __builtin_bit_cast(u32x4, qf[s][rh]);
        w ^= (u32x4)(0x80008000u);
        qf[s][rh] = __builtin_bit_cast(v8, w);
      }
  }

  f32x4 oacc[NDB][RH];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int rh = 0; rh < RH; ++rh) oacc[i][rh] = (f32x4)(0.f);
  float m_run[RH], l_run[RH];  // running row max (log2 domain, the same value in the row's 4 lanes); this lane's share of the row sum
#pragma unroll
  for (int rh = 0; rh < RH; ++rh) {
    m_run[rh] = -INFINITY;
    l_run[rh] = 0.f;
  }

  // ---- per-lane fragment addresses INSIDE THE CURRENT IMAGE (moved by +- TILE at the end of every step)
  // K fragment of step s = KV q + i, key block kb: kaddr[i] + KVB q + kb * 16 * RB: lane (n, c) reads key 16 kb + n, slot (4 s + c) ^ swizzle(key)
  FFPA_LDS const char* kaddr[KV];
#pragma unroll
  for (int i = 0; i < KV; ++i) kaddr[i] = Kt + n16 * RB + (((4 * i + c) ^ m16_k_swizzle<D>(n16)) << 4);
  // V^T fragment of column block db = VV q + i, key step ks: lane L = lane % 16 of group c reads key 32 ks + 4 c + L / 4 (+ 16 for the second
  // read), 4 columns 16 db + 4 (L % 4) ..: vaddr[i] + VVB q + (ks * 32 + {0, 16}) * RB
  FFPA_LDS const char* vaddr[VV];
  {
    const int vkey = 4 * c + (n16 >> 2);
    const int sw = m16_v_swizzle<D>(vkey);
#pragma unroll
    for (int i = 0; i < VV; ++i) vaddr[i] = Vt + vkey * RB + (((2 * i + ((n16 & 3) >> 1)) ^ sw) << 4) + 8 * (n16 & 1);
  }

  // prologue: both images of the first tile (awaited at the top of the first step)
  if (nt > t0) static_for<NP>([&](auto tc) { issue_piece(tc, t0 * BC, 0u); });

  int par = 0;  // which pair of images holds tile j
  for (int j = t0; j < nt; ++j) {
    const int k0 = j * BC;
    // the step's ONE barrier: tile j (issued a whole step ago) has landed on every wave; every wave is done with tile j - 1, whose images take tile j + 1
    dma_wait_all();
    __syncthreads();
    const uint32_t nxt = (uint32_t)__builtin_amdgcn_readfirstlane(par ? 0 : TILE);

    // ================= S^T = K.Q^T =================
    f32x4 sacc[NKB][RH];
    {
      v8 kf[N1];
      auto k_frag = [&](int n) -> v8 {
        const int s = n / NKB, kb = n % NKB;
        return *(FFPA_LDS const v8*)(kaddr[s % KV] + (s / KV) * KVB + kb * 16 * RB);
      };
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < PF1 && n < N1; ++n) kf[n] = k_frag(n);
      static_for<N1>([&](auto ic) {
        constexpr int n = decltype(ic)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (n + PF1 < N1) kf[n + PF1] = k_frag(n + PF1);
        constexpr int s = n / NKB, kb = n % NKB;
        constexpr int t = m16_piece_at(n, N1, cntQ, 0);
        if constexpr (t >= 0) {
          issue_piece_on(std::integral_constant<int, t>{}, k0 + BC, nxt, std::integral_constant<int, s == 0 ? 0 : 1>{}, sacc[kb][0], kf[n], qf[s][0]);
        } else {
          if constexpr (s == 0) M::first(sacc[kb][0], kf[n], qf[s][0]);
          else M::acc(sacc[kb][0], kf[n], qf[s][0]);
        }
#pragma unroll
        for (int rh = 1; rh < RH; ++rh) {
          if constexpr (s == 0) M::first(sacc[kb][rh], kf[n], qf[s][rh]);
          else M::acc(sacc[kb][rh], kf[n], qf[s][rh]);
        }
      });
      // MFMA result -> VALU reader wait states (invisible to the compiler inside asm); every accumulator is named so that no read of one can be
      // scheduled ahead of the statement
      static_assert(NKB == 2 || NKB == 4, "the wait-state statements below name NKB RH accumulators");
      if constexpr (NKB == 4) {
#pragma unroll
        for (int rh = 1; rh < RH; ++rh) asm volatile("" : "+v"(sacc[0][rh]), "+v"(sacc[1][rh]), "+v"(sacc[2][rh]), "+v"(sacc[3][rh]));
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(sacc[0][0]), "+v"(sacc[1][0]), "+v"(sacc[2][0]), "+v"(sacc[3][0]));
      } else {
#pragma unroll
        for (int rh = 1; rh < RH; ++rh) asm volatile("" : "+v"(sacc[0][rh]), "+v"(sacc[1][rh]));
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(sacc[0][0]), "+v"(sacc[1][0]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    // x[kb][rh][r] = score(row 16 rh + n, key k0 + 16 kb + 4 c + r) / sc: the softmax scale is folded into the exponent's FMA and applied to the row
    // max after its reduction (the kernel only ever sees sc > 0: FwdArgs.q_mode)
    float x[NKB][RH][4];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int rh = 0; rh < RH; ++rh)
#pragma unroll
        for (int r = 0; r < 4; ++r) x[kb][rh][r] = sacc[kb][rh][r];

    if constexpr (MK == 2) {
      // boolean mask bytes (non-zero = visible), straight from the caller's tensor; the lane's 4 keys of a block are consecutive
      const bool mask_free = k0 >= free_lo && k0 + BC <= free_hi;  // wave-uniform: the step lies in the mask's neutral interior (kv_bounds)
      if (a.bias_dtype == 4 && !mask_free) {
        const uint8_t* mp = (const uint8_t*)a.bias + b * a.sbias[0] + hq * a.sbias[1];
#pragma unroll
        for (int rh = 0; rh < RH; ++rh) {
          const uint8_t* mr = mp + (int64_t)qrow_c[rh] * a.sbias[2];
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
      for (int rh = 0; rh < RH; ++rh) {
        const int crow = a.causal_row_mod ? qrow[rh] % a.causal_row_mod : qrow[rh];
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

    // ================= online softmax (prefill.cuh:671-870, log2 domain; lazy rescale, threshold FwdArgs.thr) =================
    float tmax[RH];
#pragma unroll
    for (int rh = 0; rh < RH; ++rh) {
      float t = x[0][rh][0];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) t = fmaxf(t, x[kb][rh][r]);
      tmax[rh] = t;
    }
    row4_reduce_n<true, RH>(tmax);
    float m_new[RH];
    bool grow[RH], any_grow = false;
#pragma unroll
    for (int rh = 0; rh < RH; ++rh) {
      tmax[rh] *= a.scale_log2;  // (exact: rounding is monotonic, fl(sc max x) = max fl(sc x) for sc > 0; -inf stays -inf)
      m_new[rh] = fmaxf(m_run[rh], tmax[rh]);
      grow[rh] = m_new[rh] > m_run[rh] + a.thr;
      any_grow = any_grow || grow[rh];
    }
    if (__any(any_grow)) {
      float alpha[RH];
#pragma unroll
      for (int rh = 0; rh < RH; ++rh) alpha[rh] = grow[rh] ? __builtin_amdgcn_exp2f(m_run[rh] - m_new[rh]) : 1.f;
      if (j > t0) {
        // rare path: O^T lives in AGPRs; scale in place through one temporary VGPR tile
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
          for (int rh = 0; rh < RH; ++rh) {
            f32x4 t = oacc[i][rh];
            asm volatile("" : "+a"(t));
            t *= alpha[rh];
            asm volatile("" : "+a"(t));
            oacc[i][rh] = t;
            __builtin_amdgcn_sched_barrier(0);
          }
      }
#pragma unroll
      for (int rh = 0; rh < RH; ++rh) {
        l_run[rh] *= alpha[rh];
        m_run[rh] = grow[rh] ? m_new[rh] : m_run[rh];
      }
    }

    // P^T fragments: contraction slot 8 c + e of key step ks <-> key 32 ks + 16 (e / 4) + 4 c + e % 4 = x[2 ks + e / 4][rh][e % 4]
    v8 pf[NKS][RH];
#pragma unroll
    for (int rh = 0; rh < RH; ++rh) {
      const float m_use = (m_run[rh] == -INFINITY) ? 0.f : m_run[rh];
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          typedef __attribute__((ext_vector_type(2))) float f32x2;
          const f32x2 xv = {x[kb][rh][r & ~1], x[kb][rh][r | 1]};
          const f32x2 av = __builtin_elementwise_fma(xv, (f32x2)(a.scale_log2), (f32x2)(-m_use));  // (v_pk_fma_f32: the same roundings as two FMAs)
          const float p = __builtin_amdgcn_exp2f(av[r & 1]);
          psum += p;  // row sum from the unrounded P (prefill.cuh:755-756)
          pf[kb >> 1][rh][4 * (kb & 1) + r] = (T)p;
        }
      l_run[rh] += psum;
    }

    // ================= O^T += V^T.P^T =================
    {
      __builtin_amdgcn_sched_barrier(0);
      v8 vf[N2];
      auto v_frag = [&](int n) -> v8 {
        const int db = n % NDB, ks = n / NDB;
        FFPA_LDS const char* vp = vaddr[db % VV] + (db / VV) * VVB + ks * 32 * RB;
        const v4 lo = E::tr_read(vp);
        const v4 hi = E::tr_read(vp + 16 * RB);
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      };
#pragma unroll
      for (int n = 0; n < PF2 && n < N2; ++n) vf[n] = v_frag(n);
      // VALU write (the P^T conversions) -> MFMA operand read wait states: no barrier separates the softmax from this loop any more
      asm volatile("s_nop 1");
      static_for<N2>([&](auto ic) {
        constexpr int n = decltype(ic)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (n + PF2 < N2) vf[n + PF2] = v_frag(n + PF2);
        constexpr int db = n % NDB, ks = n / NDB;
        constexpr int t = cntP > 0 ? m16_piece_at(n, N2, cntP, stepP) : -1;
        if constexpr (t >= 0) issue_piece_on(std::integral_constant<int, cntQ + t>{}, k0 + BC, nxt, std::integral_constant<int, 2>{}, oacc[db][0], vf[n], pf[ks][0]);
        else M::acc_a(oacc[db][0], vf[n], pf[ks][0]);
#pragma unroll
        for (int rh = 1; rh < RH; ++rh) M::acc_a(oacc[db][rh], vf[n], pf[ks][rh]);
      });
      __builtin_amdgcn_sched_barrier(0);
    }

    // the fragment addresses follow the images
    {
      const int delta = par ? -TILE : TILE;
#pragma unroll
      for (int i = 0; i < KV; ++i) kaddr[i] += delta;
#pragma unroll
      for (int i = 0; i < VV; ++i) vaddr[i] += delta;
      par ^= 1;
    }
  }
  dma_wait_all();  // (the zero-filled pieces of the tile past the last one land before the workgroup's LDS is given to anyone else)

  // ================= epilogue (prefill.cuh:1018-1093) =================
  asm volatile("s_nop 15\n\ts_nop 3");  // last PV MFMA (inline asm) -> accumulator reads below: wait states the compiler cannot see
  float l_tot[RH], inv[RH];
#pragma unroll
  for (int rh = 0; rh < RH; ++rh) l_tot[rh] = l_run[rh];
  row4_reduce_n<false, RH>(l_tot);
#pragma unroll
  for (int rh = 0; rh < RH; ++rh) inv[rh] = __builtin_amdgcn_rcpf(l_tot[rh]);  // fully masked row: 0 * inf = NaN, as SDPA
  if (a.nsplit > 1) {
    // split-KV partial: normalised fp32 O and its LSE (merged by ffpa_fwd_merge_kernel)
#pragma unroll
    for (int rh = 0; rh < RH; ++rh) {
      if (qrow[rh] >= a.Nq) continue;
      const bool dead = !(l_tot[rh] > 0.f);
      const int64_t prow = (((int64_t)split * a.B + b) * a.Hq + hq) * a.Nq + qrow[rh];
      float* wp = a.ws_o + prow * D + 4 * c;
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
        f32x4 w;
#pragma unroll
        for (int t = 0; t < 4; ++t) w[t] = dead ? 0.f : oacc[db][rh][t] * inv[rh];
        *(f32x4*)(wp + db * 16) = w;
      }
      if (c == 0) a.ws_lse[prow] = dead ? -INFINITY : __builtin_fmaf(m_run[rh], 0.6931471805599453f, __logf(l_tot[rh]));
    }
    return;
  }
  // A lane owns 4 consecutive columns of RH rows.  Row halves in pairs (2 i, 2 i + 1): lanes c (even) and c + 1 trade one 4-column group per
  // block (v_permlane16_swap: odd 16-lane rows of the first operand <-> even rows of the second), so that the even lane stores columns
  // 16 db + 4 c .. + 8 of row half 2 i and the odd one the same columns of row half 2 i + 1 — whole 16-byte runs of ONE row.
#pragma unroll
  for (int rp = 0; rp + 1 < RH; rp += 2) {
    const int rsel = c & 1;
    const int orow = rsel ? qrow[rp + 1] : qrow[rp];
    T* op = (T*)a.o + b * a.so[0] + hq * a.so[1] + (int64_t)(orow < a.Nq ? orow : 0) * a.so[2] + 4 * (c & ~1);
    const bool ok = orow < a.Nq;
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
      v4 g0, g1;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        g0[t] = (T)(oacc[db][rp][t] * inv[rp]);
        g1[t] = (T)(oacc[db][rp + 1][t] * inv[rp + 1]);
      }
      const u32x2 x0 = __builtin_bit_cast(u32x2, g0), x1 = __builtin_bit_cast(u32x2, g1);
      u32x4 run;
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const auto sw = __builtin_amdgcn_permlane16_swap(x0[w], x1[w], false, false);
        run[w] = sw[0];
        run[2 + w] = sw[1];
      }
      if (ok && db * 16 + 4 * (c & ~1) < a.d_valid) *(u32x4*)(op + db * 16) = run;
    }
  }
  if constexpr (RH % 2 == 1) {
    // the odd row half: column blocks in pairs (db, db + 1) — the even lane keeps block db (its own group + the odd neighbour's), the odd
    // lane block db + 1 (the even neighbour's group + its own): 16-byte runs again
    static_assert(NDB % 2 == 0, "column blocks in pairs");
    constexpr int rl = RH - 1;
    const int rsel = c & 1;
    const int orow = qrow[rl];
    T* op = (T*)a.o + b * a.so[0] + hq * a.so[1] + (int64_t)(orow < a.Nq ? orow : 0) * a.so[2] + 4 * (c & ~1) + 16 * rsel;
    const bool ok = orow < a.Nq;
#pragma unroll
    for (int db = 0; db < NDB; db += 2) {
      v4 g0, g1;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        g0[t] = (T)(oacc[db][rl][t] * inv[rl]);
        g1[t] = (T)(oacc[db + 1][rl][t] * inv[rl]);
      }
      const u32x2 x0 = __builtin_bit_cast(u32x2, g0), x1 = __builtin_bit_cast(u32x2, g1);
      u32x4 run;
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const auto sw = __builtin_amdgcn_permlane16_swap(x0[w], x1[w], false, false);
        run[w] = sw[0];
        run[2 + w] = sw[1];
      }
      if (ok && (db + rsel) * 16 + 4 * (c & ~1) < a.d_valid) *(u32x4*)(op + db * 16) = run;
    }
  }
  if (a.lse != nullptr && c == 0) {
#pragma unroll
    for (int rh = 0; rh < RH; ++rh)
      if (qrow[rh] < a.Nq) a.lse[((int64_t)b * a.Hq + hq) * a.Nq + qrow[rh]] = __builtin_fmaf(m_run[rh], 0.6931471805599453f, __logf(l_tot[rh]));
  }
}

}  // namespace ffpa
