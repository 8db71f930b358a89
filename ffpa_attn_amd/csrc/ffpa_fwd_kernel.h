// ffpa_fwd_kernel.h — Split-D fused attention forward for gfx950 (MI355X, CDNA4).
//
// Written from scratch for wave64 / MFMA / 160 KiB LDS / 512-register SIMDs.  The
// algorithm (what is computed, in which order) follows the reference's native kernel
// split_d_fwd_sm80 (csrc/cuffpa/native/sm_80/split_d.cuh:96-777) and its helpers
// (csrc/cuffpa/native/prefill.cuh:671-1093); the mapping onto the machine does not.
//
// Machine mapping (see DESIGN.md for the derivation and the measurements behind each choice):
//   * one workgroup = 4 waves = one wave per SIMD, each wave owns the SIMD's whole 512-entry register file.
//     A wave owns 32 query rows and DW = D/ND output columns:
//       ND = 1 (D <= 512):  4 waves x 32 rows            -> BR = 128 rows / workgroup, BC = 128 keys / tile for
//                                                           D <= 320, 64 above
//       ND = 2 (D  > 512):  2 row blocks x 2 D-halves    -> BR =  64, BC = 32
//       ND = 4 (Nq <= 32):  1 row block  x 4 D-quarters  -> BR =  32, BC = 32 (short-query / decode launches,
//                                                           with the KV axis split over workgroups)
//   * S^T = K.Q^T  (v_mfma_f32_32x32x16, A = K rows from LDS, B = Q rows from VGPRs): every lane then owns ONE
//     query column, so the softmax row reductions are in-lane plus a single lane^32 exchange, and the C layout
//     of S^T is already the B-operand layout of the second contraction (P never leaves registers).  MFMA row a
//     is fed key pi(a), chosen so that each lane holds 16 CONSECUTIVE keys of every 32-key block.
//   * O^T += V^T.P^T (A = V^T via ds_read_b64_tr_b16 from a row-major V tile in LDS, B = P^T from registers,
//     accumulator O^T = 16*DW/32 AGPRs per lane).
//   * Split-D: the Q.K contraction walks D in 16-wide steps against Q fragments that stay resident in VGPRs
//     (DW/4 registers); for ND > 1 each wave contracts only its own slice of D and the partial S^T tiles are
//     summed through LDS.  The O^T accumulator is split over D the same way, so registers + LDS stay bounded
//     in D.
//   * K and V tiles ([BC keys][D]) are brought in by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR round
//     trip), issued through inline asm between the MFMAs; bank-conflict-avoiding XOR swizzles are applied on
//     the per-lane SOURCE offset because the DMA destination is lane-linear; rows past the last key are
//     zero-filled by the buffer descriptor's range check.  Per KV tile (ND <= 2; the ND = 4 tiles use bursts): QK^T(j) with the
//     V(j) pieces interleaved -> barrier A1 -> first K(j+1) pieces -> softmax -> counted vmcnt + barrier A2 ->
//     PV(j) with the remaining K(j+1) pieces interleaved -> barrier B.
//   * blockIdx is remapped so that all row tiles of one (batch, head) run on the same XCD at the same time
//     and share K/V through that XCD's L2.
//
// ffpa_fwd_m16_kernel.h holds a second mapping of the same tiles and pipeline onto the 16x16x32 MFMA shape (it sustains a higher
// power-capped rate on this chip): EVERY prefill launch at head dims >= 128 runs there (ffpa_fwd_m16w_kernel.h: its wide-row tile for D = 320).
// What this file's kernel still serves, and why its prefill tiles (ND = 1 / 2) stay:
//   * the short-query / decode tiles of every head dim (ND = 4 / 2, KV axis split over workgroups), the split-KV merge and the mask-range scan;
//   * the PRODUCT prefill launches of D = 64 only — measured: the 16x16x32 mapping is 5 % slower there (899 vs 855 TFLOPS, profiles/r03_m16_small_d.txt:
//     a 64-wide head dim is two contraction steps of the 16x16x32 shape, its fragment sharing has nothing to amortise);
//   * the TEST-ONLY register-staged twins (SAFE = true; libffpa_attn_hip_test.so, D = 64 / 128 / 320 / 512 / 640 / 1024): the same recurrence on
//     another mapping of the matrix core and another data path (no LDS-DMA, no transpose reads) — what tests/test_m16_gpu.py bisects the
//     product kernels against.  They are why the ND = 2 prefill tile and the 128-key tile are still here although no product launch reaches them.
#pragma once

#include "ffpa_common.h"
#include "ffpa_philox.h"

// ---- build-time tunables (defaults = the shipped configuration; tools/gpu_ab.py builds variants) ----
#ifndef FFPA_PF1
#define FFPA_PF1 6  // QK: LDS reads run this many MFMAs ahead of their consumer
#endif
#ifndef FFPA_PF2
#define FFPA_PF2 4  // PV: ditto (two transpose reads per MFMA)
#endif
#ifndef FFPA_DMA_STEP
#define FFPA_DMA_STEP 4  // interleaved mode: one DMA piece every this many MFMAs (ND == 1): spreads the tile evenly over the loop
#endif
#ifndef FFPA_DMA_STEP_ND2
#define FFPA_DMA_STEP_ND2 2  // ditto for the split-D kernels (+2 % over 1)
#endif
#ifndef FFPA_HOIST_MAX_D
#define FFPA_HOIST_MAX_D 448      // ND == 1: hoist the per-lane DMA source offsets up to this head dim
#endif
#ifndef FFPA_HOIST_ND2_MAX_D
#define FFPA_HOIST_ND2_MAX_D 960  // ND == 2 (split-D): every head dim whose rows are not whole pieces
#endif
#ifndef FFPA_BC128_MAX_D
#define FFPA_BC128_MAX_D 320  // ND == 1 head dims up to this use 128-key tiles (2*128*D*2 B of LDS <= 160 KiB)
#endif
#ifndef FFPA_K_PRE
#define FFPA_K_PRE 8  // interleaved mode: this many K(j+1) pieces are issued right after QK^T (they stream under
#endif                //   the softmax); V(j) is then awaited with a counted vmcnt just before PV.  0 = one barrier A
#ifndef FFPA_K_PRE_ND2
#define FFPA_K_PRE_ND2 64  // split-D kernels: as FFPA_K_PRE; it is clamped to the tile's pieces (a multiple of 4)
#endif
// Settled schedule choices that used to be on / off switches here (each measured, tools/experiments/r05_pruned_switches_2.diff restores them): DMA pieces go out
// between the MFMAs of the prefill tiles (ND <= 2; the short-query ND = 4 tiles keep bursts); the early K(j+1) pieces are spread over the softmax stages in four
// groups (+ 0.2 ... 1.4 % at D <= 512, + 7 % at D = 1024); tile-invariant per-lane DMA source offsets are hoisted where the registers allow; D % 512 == 0 rows use the
// wave-uniform row form (3 scalar instructions per piece); the first PV fragments are read right behind barrier A; QK^T runs d-step outer, PV key-step outer; the 16-bit
// bias tile of the split-D tiles is loaded ahead of barrier A; the short-query builds carry a non-temporal form of their K / V DMA (FwdArgs.flags picks per launch).
namespace ffpa {

// Additive bias for the 16 scores one lane holds of a 32-key block:
// x[r] += bias[row][k0 + (r&3) + 8 (r>>2) + 4 h] * log2(e)   (prefill.cuh:556-658; here the
// bias is added after the scale instead of being pre-divided by it — same value).
template <typename BT>
__device__ __forceinline__ void add_bias_block(float (&x)[16], const void* bias, int64_t row_off,
                                               int64_t stride_key, int key_base, int nkv) {
  const BT* bp = (const BT*)bias + row_off;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int key = key_base + r;
    key = key < nkv ? key : nkv - 1;
    x[r] += (float)bp[key * stride_key] * 1.4426950408889634f;
  }
}

// Vector variant for unit key stride: the lane's 16 scores of a key block are 16 CONSECUTIVE keys (the key <->
// MFMA-row map of the kernel is chosen for exactly that), read as 16 / W loads of W elements.
template <typename BT, int W>
__device__ __forceinline__ void add_bias_block_vec(float (&x)[16], const void* bias, int64_t row_off, int key_base) {
  typedef __attribute__((ext_vector_type(W))) BT bvec;
  const BT* bp = (const BT*)bias + row_off + key_base;
  bvec raw[16 / W];
#pragma unroll
  for (int i = 0; i < 16 / W; ++i) raw[i] = *(const bvec*)(bp + W * i);
#pragma unroll
  for (int i = 0; i < 16 / W; ++i)
#pragma unroll
    for (int t = 0; t < W; ++t) x[W * i + t] += (float)raw[i][t] * 1.4426950408889634f;
}

// Boolean mask (FFPA_BIAS_BOOL8: one byte per score, non-zero = visible): what the reference's host turns into an additive
// 0 / -inf tensor in q.dtype before the launch (functional.py:891-898) is applied here straight from the caller's bytes —
// x += 0 or x = -inf are the same scores, without the mask-sized temporary.  16 consecutive keys per lane and key block.
__device__ __forceinline__ void apply_bool_block(float (&x)[16], const void* mask, int64_t row_off, int64_t stride_key, int key_base, int nkv) {
  const uint8_t* mp = (const uint8_t*)mask + row_off;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int key = key_base + r;
    key = key < nkv ? key : nkv - 1;
    if (mp[key * stride_key] == 0) x[r] = -INFINITY;
  }
}
__device__ __forceinline__ void apply_bool_block_vec(float (&x)[16], const void* mask, int64_t row_off, int key_base) {
  const u32x4 raw = *(const u32x4*)((const uint8_t*)mask + row_off + key_base);
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if (((raw[r >> 2] >> (8 * (r & 3))) & 0xffu) == 0u) x[r] = -INFINITY;
}

// ---------------------------------------------------------------------------------
// KV-split launches: merging the partials inside the launch (FwdArgs.tickets != NULL).
//
// Every split workgroup of a row tile calls this after storing its normalised fp32 partial + LSE to the workspace.  Hand-off in the
// counter form of the cdna guide's Guideline 16: every wave drains its stores -> workgroup barrier -> lane 0: agent-scope release fence,
// the post-write-back wait restated in asm (ROCm 7.2 drops the compiler's own when the wave's scoreboard is empty), ONE relaxed
// agent-scope fetch_add on the tile's ticket.  The workgroup that draws nsplit - 1 is the last to arrive: lane 0 takes ONE agent-scope
// acquire (invalidates this CU's L1: the other splits' partials were written by other CUs, possibly on other XCDs), barrier, then all
// waves read the partials with plain loads and combine them exactly as ffpa_fwd_merge_kernel does (O = sum_s w_s O_s / sum_s w_s,
// w_s = exp(LSE_s - max LSE), LSE = max + ln sum w_s: csrc/cuffpa/native/sm_80/split_kv.cuh:329-455), and the ticket goes back to
// zero for the next launch.  Placement-independent: nothing assumes which CU / XCD ran which split, or in which order.
// `scratch`: >= 16 + 4 * kMergeMaxSplits * 4 bytes of LDS nobody else uses any more (the K / V tile area after the tile loop).
// ---------------------------------------------------------------------------------
constexpr int kMergeMaxSplits = 1024;
// 16-byte write-through store (sc0 sc1: the line goes to memory, not just to this XCD's L2): a partial stored this way needs no
// release fence (buffer_wbl2: 2 - 6 us per workgroup) before the ticket, only the wave's own vmcnt(0) (Guideline 16, form R1).
__device__ __forceinline__ void store_write_through(float* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
// kWriteThrough: the caller stored its partial with store_write_through / agent-scope atomic stores.
template <typename T, bool kWriteThrough = false>
__device__ __forceinline__ void split_arrive_and_merge(const FwdArgs& a, int D, int tile_id, int b, int hq, int row0, int nrows, FFPA_LDS char* scratch) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's partial stores have left the CU
  __syncthreads();
  FFPA_LDS int* flag = (FFPA_LDS int*)scratch;
  if (tid == 0) {
    if constexpr (!kWriteThrough) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int t = __hip_atomic_fetch_add(a.tickets + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == a.nsplit - 1) ? 1 : 0;
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *flag = last;
  }
  __syncthreads();
  if (*flag == 0) return;
  FFPA_LDS float* wsh = (FFPA_LDS float*)(scratch + 16) + wave * kMergeMaxSplits;
  const int64_t rows = (int64_t)a.B * a.Hq * a.Nq;
  const int64_t sstride = rows * D;
  for (int r0 = 0; r0 < nrows; r0 += 4) {  // one row per wave and round (uniform trip count: the barriers below are workgroup barriers)
    const int qrow = row0 + r0 + wave;
    const bool act = r0 + wave < nrows && qrow < a.Nq;
    const int64_t row = ((int64_t)b * a.Hq + hq) * a.Nq + (act ? qrow : 0);
    float mx = -INFINITY;
    if (act)
      for (int s = lane; s < a.nsplit; s += 64) mx = fmaxf(mx, a.ws_lse[s * rows + row]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float wsum = 0.f;
    if (act)
      for (int s = lane; s < a.nsplit; s += 64) {
        const float w = (mx == -INFINITY) ? 0.f : __expf(a.ws_lse[s * rows + row] - mx);
        wsh[s] = w;
        wsum += w;
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wsum += __shfl_xor(wsum, o);
    __syncthreads();  // the weights are visible to the whole wave
    if (act) {
      const float inv = 1.f / wsum;  // every share empty -> 0 * inf = NaN, like an unsplit fully masked row
      T* op = (T*)a.o + b * a.so[0] + hq * a.so[1] + (int64_t)qrow * a.so[2];
      for (int d = lane * 4; d < a.d_valid; d += 256) {
        const float* src = a.ws_o + row * D + d;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 8 <= a.nsplit; s += 8) {
          f32x4 t[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) t[u] = *(const f32x4*)(src + (s + u) * sstride);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float w = wsh[s + u];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += w * t[u][e];
          }
        }
        for (; s < a.nsplit; ++s) {
          const f32x4 t = *(const f32x4*)(src + s * sstride);
          const float w = wsh[s];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += w * t[e];
        }
        typename Elem<T>::v4 w4;
#pragma unroll
        for (int e = 0; e < 4; ++e) w4[e] = (T)(acc[e] * inv);
        *(typename Elem<T>::v4*)(op + d) = w4;
      }
      if (a.lse != nullptr && lane == 0) a.lse[row] = (mx == -INFINITY) ? -INFINITY : mx + __logf(wsum);
    }
    __syncthreads();  // the weights are free for the next round
  }
  if (tid == 0) a.tickets[tile_id] = 0;  // (the next launch that uses these tickets is ordered behind this one by its stream)
}

// DROP selects the dropout-capable build of the kernel: kept out of the default instantiation because its
// Philox temporaries push hipcc into spilling Q fragments inside the QK^T loop (and every reload drains the
// DMA queue); dropout launches pay that, plain launches do not.
// BTILE selects the build that stages 16-bit bias tiles through LDS (FwdArgs.bias_tile): its own instantiation for the same
// reason as DROP — inside the default kernel its descriptor arithmetic costs the hot loops scalar registers they do not have.
// MK (mask kind) selects which bias / mask paths a build carries: 0 = none — the build for calls without any attn_bias / mask
// ranges (is_causal and ragged tails are structural and stay): with every bias path compiled out the prefill kernels lose the
// scalar registers and the per-tile branch chain those paths cost them, + 1.5 ... 4 % (D = 512: 1215 vs 1187 TFLOPS,
// D = 320: 1170 vs 1131, N = 2048: + 4 %, measured A/B); 2 = boolean masks only (bytes + mask ranges; what
// ffpa_attn_func(attn_mask=<bool>) launches: config 4 as specified); 1 = every path.
template <typename T, int D, int ND, bool SAFE, bool DROP = false, bool BTILE = false, int MK = 1>
__global__ __launch_bounds__(256) void ffpa_fwd_split_d_kernel(const FwdArgs a) {
  constexpr bool MASK = MK != 0;
  constexpr bool kBoolOnly = MK == 2;
  using E = Elem<T>;
  using v8 = typename E::v8;
  using v4 = typename E::v4;
  static_assert(ND == 1 || ND == 2 || ND == 4, "D is split over 1, 2 or all 4 waves");
  static_assert(D % 64 == 0, "head dim must be a multiple of 64");
  constexpr int DW = D / ND;       // output columns owned by one wave
  constexpr int NDB = DW / 32;     // 32-column O^T blocks per wave
  constexpr int KS = DW / 16;      // QK contraction steps per wave
  // keys per tile: 128 for small head dims (fewer barriers / phase fills per key), 64 up to D = 512, 32 when D
  // is split over waves (ND == 4, short-query launches: small tiles, 2 workgroups / CU)
  constexpr int BC = (ND == 1) ? ((D <= FFPA_BC128_MAX_D && !BTILE) ? 128 : 64) : splitd_block_keys(D, ND);  // (bias-tile builds: 64 keys, the LDS must hold the bias tiles too)
  constexpr int NKB = BC / 32;     // 32-key S^T blocks per tile
  constexpr int NKS = BC / 16;     // PV contraction steps per tile
  constexpr int NQB = 4 / ND;      // 32-row blocks per workgroup
  constexpr int BR = 32 * NQB;
  constexpr int RB = D * 2;        // tile row bytes
  constexpr int TILE = BC * RB;
  constexpr int PF1 = FFPA_PF1;
  constexpr int PF2 = FFPA_PF2;
  constexpr int PPW = BC * D * 2 / 4096;  // 1 KiB DMA pieces per wave per tile
  constexpr bool kInterleave = ND <= 2;  // LDS-DMA pieces go out between the MFMAs; the short-query (ND = 4) tiles keep bursts behind the barriers
  constexpr int kStep = (ND == 1) ? FFPA_DMA_STEP : (ND == 2) ? FFPA_DMA_STEP_ND2 : 1;  // MFMAs between two DMA pieces
  static_assert(!kInterleave || PPW * kStep <= (DW / 16) * (BC / 32), "DMA pieces must fit the QK loop");
  // Head dims whose rows are not a whole number of 1 KiB pieces need ~12 VALU instructions per piece for the
  // per-lane source offset (constant division, swizzle).  The offsets are tile-invariant: where the register
  // budget allows they are hoisted into PPW + PPW VGPRs.
  constexpr bool kPad = SAFE;  // s_nop 1 in front of the asm MFMAs (see FFPA_MFMA_PAD)
  constexpr bool kRowUniform = (D * 2) % 1024 == 0;
  constexpr bool kHoist = !kRowUniform && !SAFE && !DROP && (ND == 1 ? D <= FFPA_HOIST_MAX_D : (ND == 2 && D <= FFPA_HOIST_ND2_MAX_D));
  // Row-uniform head dims: wave w stages keys 16 a + 4 w + b (a < BC/16, b < 4) so that only four K and
  // four V swizzled lane offsets exist (K: slot ^ (4 w + b); V: slot ^ 4 b) and live in 8 VGPRs.
  constexpr bool kRowDma = kRowUniform && !SAFE && kInterleave;  // (burst-mode kernels keep stage_piece)
  constexpr int RPP = kRowUniform ? RB / 1024 : 1;  // pieces per row
  constexpr int KPW = BC / 4;                       // keys staged per wave per tile
  static_assert(!kRowDma || (D % 128 == 0 && KPW * RPP == PPW && BC % 16 == 0), "row DMA layout");
  constexpr int kPreReq = (ND == 2) ? FFPA_K_PRE_ND2 : FFPA_K_PRE;
  constexpr bool kSpreadReq = !SAFE;  // the early K(j+1) pieces go out in four groups between the softmax stages (split-D tiles: all of K(j+1) does)
  constexpr int kPre = !(kInterleave && kPreReq > 0) ? 0
                       : kSpreadReq              ? ((kPreReq < PPW ? kPreReq : PPW) / 4) * 4
                                                 : (kPreReq <= PPW ? kPreReq : 0);

  extern __shared__ __attribute__((aligned(16))) char smem[];
  FFPA_LDS char* const Kt = (FFPA_LDS char*)smem;
  FFPA_LDS char* const Vt = Kt + TILE;
  FFPA_LDS char* const Xb = Kt + 2 * TILE;  // ND == 2: partial-S exchange, 4 KiB per wave
  FFPA_LDS char* const Bl = Kt + 2 * TILE + splitd_exchange_bytes(D, ND);  // key-bias row cache (FwdArgs.bias_lds bytes, when enabled)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31;
  const int h = lane >> 5;
  const int qb = wave / ND;  // row block of this wave
  const int dh = wave % ND;  // which D/ND slice of the head dim it owns

  // ---- workgroup -> (batch, head, row tile).  Block b runs on XCD b % 8; give every
  // XCD a contiguous range of virtual ids so that the row tiles of one head (which
  // stream the same K/V) are co-resident on one XCD and hit in its L2.
  int vid = blockIdx.x;
  if (!(a.flags & kFlagNoXcdRemap)) vid = xcd_logical_id(vid, gridDim.x, a.xcd_group);
  const int split = vid % a.nsplit;
  vid /= a.nsplit;
  const int bh = vid / a.nqt;
  int qt = vid - bh * a.nqt;
  if (a.causal || (MASK && a.kv_bounds != nullptr)) qt = a.nqt - 1 - qt;  // longest rows first (mask ranges: their usual source is a causal-like mask)
  const int b = bh / a.Hq;
  const int hq = bh - b * a.Hq;
  const int hkv = hq / a.group;

  const int q0 = qt * BR;
  const int wq0 = q0 + qb * 32;
  const int qrow = wq0 + l31;
  const int qrow_c = qrow < a.Nq ? qrow : a.Nq - 1;

  const T* __restrict__ Kg = (const T*)a.k + b * a.sk[0] + hkv * a.sk[1];
  const T* __restrict__ Vg = (const T*)a.v + b * a.sv[0] + hkv * a.sv[1];
  const uint32_t k_row_bytes = (uint32_t)a.sk[2] * 2u;
  const uint32_t v_row_bytes = (uint32_t)a.sv[2] * 2u;

  // the caller's head dim when it is not a multiple of 64 (else == D): bytes / 16-byte slots of a row that hold data
  const uint32_t rb_valid = (uint32_t)a.d_valid * 2u;
  const int slots_valid = a.d_valid >> 3;

  // ---- bias tiles through LDS (FwdArgs.bias_tile): the wave's private [32 rows][BC keys] 16-bit image, row stride BC * 2 bytes,
  // 16-byte slot s of row r stored at slot s ^ g(r) (source-side swizzle, like K / V) so that the 16-lane groups of the
  // ds_read_b128 that fetch one slot of 16 different rows are conflict-free.  Filled by BC / 16 pieces of 1 KiB.
  constexpr bool kBiasTile = BTILE && MASK && ND <= 2 && !SAFE && BC <= 64;
  constexpr int kBtSlots = BC / 8;              // 16-byte slots per bias row
  constexpr int kBtRowsPerPiece = 64 / kBtSlots;  // rows one 1 KiB piece covers
  constexpr int kBtPieces = 32 / kBtRowsPerPiece;
  FFPA_LDS char* const Bt = Bl + wave * (32 * BC * 2);
  auto bias_swz = [](int row) { return BC == 64 ? (row >> 1) & 7 : (row >> 2) & 3; };
  uint32_t bvo[kBiasTile ? (BC == 64 ? 2 : 1) : 1];  // per-lane source offsets (piece parity: the swizzle of BC = 64 depends on it)
  if constexpr (kBiasTile) {
    const uint32_t rs = (uint32_t)a.sbias[2] * 2u;
    const int r = lane / kBtSlots, sl = lane % kBtSlots;
    bvo[0] = (uint32_t)r * rs + (uint32_t)((sl ^ bias_swz(r)) << 4);
    if constexpr (BC == 64) bvo[1] = (uint32_t)r * rs + (uint32_t)((sl ^ bias_swz(kBtRowsPerPiece + r)) << 4);
  }
  // piece i of the bias tile of KV step key0 for this wave's rows (issued one step ahead, awaited by barrier B's drain)
  auto issue_bias = [&](auto ic, int key0) {
    if constexpr (kBiasTile) {
      constexpr int i = decltype(ic)::value;
      const uint32_t rs = (uint32_t)a.sbias[2] * 2u;
      const int64_t plane = 2 * (b * a.sbias[0] + hq * a.sbias[1]);
      const int64_t first = (int64_t)wq0 * rs + 2 * (int64_t)key0;                           // this wave's first row, first key of the step
      const int64_t plane_bytes = (int64_t)(a.Nq - 1) * rs + 2 * (int64_t)a.Nkv;             // rows past Nq / the plane's end read as zeros
      int64_t left = plane_bytes - first;
      left = left < 0 ? 0 : (left > 0x7fffffff ? 0x7fffffff : left);
      const u32x4 rsrc = make_rsrc((const char*)a.bias + plane + first, (uint32_t)left);
      lds_dma_16(rsrc, (uint32_t)(uintptr_t)(Bt + i * 1024), bvo[BC == 64 ? (i & 1) : 0], (uint32_t)(i * kBtRowsPerPiece) * rs);
    }
  };

  // ---- DMA issue helpers (piece i of this wave for the tile starting at key0)
  uint32_t krel[kHoist ? BC * D * 2 / 4096 : 1], vrel[kHoist ? BC * D * 2 / 4096 : 1];
  if constexpr (kHoist) {
    constexpr int SPRh = D / 8, PPWh = BC * D * 2 / 4096;
#pragma unroll
    for (int i = 0; i < PPWh; ++i) {
      const int g = (wave * PPWh + i) * 64 + lane;
      const int key = g / SPRh;
      const int slot = g - key * SPRh;
      krel[i] = (uint32_t)key * k_row_bytes + (uint32_t)((slot ^ k_slot_swizzle<D>(key)) << 4);
      if ((slot ^ k_slot_swizzle<D>(key)) >= slots_valid) krel[i] = kDmaOob;  // K columns past the head dim read as zeros
      vrel[i] = (uint32_t)key * v_row_bytes + (uint32_t)((slot ^ v_slot_swizzle<D>(key)) << 4);
    }
  }
  uint32_t kvo[kRowDma ? 4 : 1], vvo[kRowDma ? 4 : 1];        // per-lane swizzled column offsets
  uint32_t kro[kRowDma ? KPW : 1], vro[kRowDma ? KPW : 1];    // scalar: row offset of staged key jk
  uint32_t k_lds = 0, v_lds = 0;
  if constexpr (kRowDma) {
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
      kvo[bb] = (uint32_t)((lane ^ (4 * wave + bb)) << 4);
      if (RPP == 1 && (lane ^ (4 * wave + bb)) >= slots_valid) kvo[bb] = kDmaOob;  // (two-piece rows: masked per piece below)
      vvo[bb] = (uint32_t)((lane ^ (4 * bb)) << 4);
    }
#pragma unroll
    for (int jk = 0; jk < KPW; ++jk) {
      const uint32_t key = (uint32_t)(16 * (jk >> 2) + 4 * wave + (jk & 3));
      kro[jk] = key * k_row_bytes;
      vro[jk] = key * v_row_bytes;
    }
    k_lds = (uint32_t)(uintptr_t)Kt + (uint32_t)(4 * wave * RB);
    v_lds = (uint32_t)(uintptr_t)Vt + (uint32_t)(4 * wave * RB);
  }
  // All three forms address the tile through tile_src(): offsets are relative to the tile's first row and the
  // descriptor zero-fills rows past the last key.
  constexpr bool kNt = ND >= 2;  // the short-query builds can stream K / V past the caches (see lds_dma_16) ...
  const bool stream_kv = kNt && (a.flags & kFlagStreamKV) != 0;  // ... when the launch side says every byte has one reader and the caches cannot hold them
  auto issue_k = [&](auto ic, int key0, int dlane) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    const TileSrc ts = tile_src<BC>(Kg, k_row_bytes, key0, a.Nkv, rb_valid);
    if constexpr (kRowDma) {
      constexpr int jk = i / RPP, half = i % RPP;
      uint32_t kv = kvo[jk & 3];
      if constexpr (RPP > 1 && half == RPP - 1) {
        // the row's last piece: lanes whose 16-byte slot lies at or past the caller's head dim fetch zeros (a wave-uniform
        // branch: launches whose head dim is the kernel's own skip the per-lane test)
        if (a.d_valid != D) {
          if (half * 64 + (dlane ^ (4 * wave + (jk & 3))) >= slots_valid) kv = kDmaOob;
        }
      }
      if (kNt && stream_kv) lds_dma_row<(16 * (jk >> 2) + (jk & 3)) * RB, half * 1024, true>(ts.rsrc, k_lds, kv, kro[jk]);
      else lds_dma_row<(16 * (jk >> 2) + (jk & 3)) * RB, half * 1024, false>(ts.rsrc, k_lds, kv, kro[jk]);
    } else if constexpr (kHoist) {
      lds_dma_16_sel<kNt>(stream_kv, ts.rsrc, (uint32_t)(uintptr_t)(Kt + (wave * PPW + i) * 1024), krel[i], 0u);
    } else {
      stage_piece<T, D, BC, false, SAFE, kNt>(ts.rsrc, ts.base, k_row_bytes, ts.rows, Kt, wave, dlane, i, slots_valid, stream_kv);
    }
  };
  auto issue_v = [&](auto ic, int key0, int dlane) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    const TileSrc ts = tile_src<BC>(Vg, v_row_bytes, key0, a.Nkv, rb_valid);
    if constexpr (kRowDma) {
      constexpr int jk = i / RPP, half = i % RPP;
      if (kNt && stream_kv) lds_dma_row<(16 * (jk >> 2) + (jk & 3)) * RB, half * 1024, true>(ts.rsrc, v_lds, vvo[jk & 3], vro[jk]);
      else lds_dma_row<(16 * (jk >> 2) + (jk & 3)) * RB, half * 1024, false>(ts.rsrc, v_lds, vvo[jk & 3], vro[jk]);
    } else if constexpr (kHoist) {
      lds_dma_16_sel<kNt>(stream_kv, ts.rsrc, (uint32_t)(uintptr_t)(Vt + (wave * PPW + i) * 1024), vrel[i], 0u);
    } else {
      stage_piece<T, D, BC, true, SAFE, kNt>(ts.rsrc, ts.base, v_row_bytes, ts.rows, Vt, wave, dlane, i, slots_valid, stream_kv);
    }
  };
  auto issue_k_tile = [&](int key0) __attribute__((always_inline)) {
    const int dl = opaque_lane(lane);
    static_for<PPW>([&](auto ic) { issue_k(ic, key0, dl); });
  };
  auto issue_v_tile = [&](int key0) __attribute__((always_inline)) {
    const int dl = opaque_lane(lane);
    static_for<PPW>([&](auto ic) { issue_v(ic, key0, dl); });
  };

  // ---- KV tile range (split_d.cuh:222-228: causal tiles past the diagonal are skipped)
  int nt = (a.Nkv + BC - 1) / BC;
  if (a.causal) {
    const int last_row = a.causal_row_mod ? a.causal_row_mod - 1 : q0 + BR - 1;
    const int64_t last = (int64_t)last_row + a.causal_offset;
    const int ntc = last < 0 ? 0 : (int)(last / BC) + 1;
    nt = nt < ntc ? nt : ntc;
  }
  int t0 = split * a.tiles_per_split;  // this workgroup's share of the KV tiles
  {
    const int t1 = t0 + a.tiles_per_split;
    nt = nt < t1 ? nt : t1;
  }
  // keys [free_lo, free_hi): the mask is neutral (adds exactly 0 / every key visible) for EVERY row of this wave's
  // 32-row block — KV tiles inside that range skip the mask loads altogether (the interior of a causal / sliding-window /
  // padding mask costs what an unmasked launch costs; only the tiles the mask's edge crosses read it)
  int free_lo = 0, free_hi = 0;
  if (MASK && a.kv_bounds != nullptr) {
    // the caller's mask leaves only keys [first, end) visible to the 32-row blocks of this row tile
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

  // ---- Q fragments: B operand of S^T = K.Q^T.  lane (row l31, half h) holds
  // Q[row][dh*DW + 16 s + 8 h .. +8] for every contraction step s.
  v8 qf[KS];
  {
    const T* qp = (const T*)a.q + b * a.sq[0] + hq * a.sq[1] + (int64_t)qrow_c * a.sq[2] + dh * DW + h * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      // columns at and past the caller's head dim are zeros (and are not read: the last row may end the allocation)
      const u32x4 z = {0u, 0u, 0u, 0u};
      qf[s] = (dh * DW + s * 16 + h * 8 < a.d_valid) ? *(const v8*)(qp + s * 16) : __builtin_bit_cast(v8, z);
    }
  }

  f32x16 oacc[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i) oacc[i] = (f32x16)(0.f);
  float m_run = -INFINITY;  // running row max, log2 domain (includes softmax_scale)
  float l_run = 0.f;        // this lane's share of the row sum (other share: lane ^ 32)

  // ---- key <-> MFMA-row map.  Row a of S^T = K.Q^T (A operand lane a) is fed key pi(a) =
  // 16 (a/4 % 2) + a%4 + 4 (a/8) of the 32-key block, so that the C layout — row (r&3) + 8 (r>>2) + 4 h in
  // register r of lane half h — leaves each lane with the 16 CONTIGUOUS keys 16 h + r: masks, bias tiles and
  // dropout counters are then plain runs of keys.  V^T below uses the same map (the key index is a contraction
  // index of the second product, so any permutation is free as long as both agree).
  // ---- per-lane LDS addresses, hoisted so that the MFMA loops carry only immediates.
  // K fragment of step s = 8 q + i, key block kb:  kaddr[i] + 256 q + kb*32*RB
  //   (slot (c0 + 2 s + h) ^ kx: adding 16 q slots commutes with a 4-bit XOR)
  FFPA_LDS const char* kaddr[8];
  {
    const int kpi = 16 * ((l31 >> 2) & 1) + (l31 & 3) + 4 * (l31 >> 3);
    const int kx = k_slot_swizzle<D>(kpi);
    const int c0 = dh * (DW / 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) kaddr[i] = Kt + kpi * RB + (((c0 + 2 * i + h) ^ kx) << 4);
  }
  // V^T fragment of column block db = 4 q + i, key step ks: contraction slot (h, e) <-> key
  // 32 (ks/2) + 16 h + 8 (ks%2) + e; two transpose reads (e < 4, e >= 4) at
  // vaddr[i] + 256 q + (32 (ks/2) + 8 (ks%2) + {0, 4}) * RB.  ds_read immediates are 16 bits: key steps whose
  // row offset would overflow use a second base, +64 rows.
  constexpr bool kVHi = NKS > 4 && (32 * ((NKS - 1) >> 1) + 12) * RB + 768 > 65535;
  FFPA_LDS const char* vaddr[kVHi ? 8 : 4];
  {
    const int j4 = (lane & 15) >> 2;
    const int vsw = v_slot_swizzle<D>(j4) * 16;
    const int vcol = dh * DW * 2 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;  // bytes
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      vaddr[i] = Vt + (16 * h + j4) * RB + ((vcol + i * 64) ^ vsw);
      if constexpr (kVHi) vaddr[4 + i] = vaddr[i] + 64 * RB;
    }
  }

  if (MASK && !kBoolOnly && a.bias_lds > 0 && nt > t0) {
    // key bias [.., .., 1, Nkv]: every row of the workgroup adds the same Nkv values — fetch them once (a few KiB) instead of
    // 2-4 latency-exposed global loads per lane and key block in every tile; bytes past Nkv are zeros (those keys get the
    // tail mask).  Made visible by the barrier below.
    const int esz = a.bias_dtype == 3 ? 4 : 2;
    const char* src = (const char*)a.bias + (int64_t)esz * (b * a.sbias[0] + hq * a.sbias[1]);
    const int valid = a.Nkv * esz;
    for (int i = tid * 16; i < a.bias_lds; i += 256 * 16) {
      u32x4 w = {0u, 0u, 0u, 0u};
      if (i < valid) w = *(const u32x4*)(src + i);
      *(FFPA_LDS u32x4*)(Bl + i) = w;
    }
  }
  if (nt > t0) {
    issue_k_tile(t0 * BC);
    if constexpr (kBiasTile) {
      static_for<kBtPieces>([&](auto ic) { issue_bias(ic, t0 * BC); });
    }
    dma_wait_all();
    __syncthreads();  // K(t0) landed and visible
    if constexpr (!kInterleave) issue_v_tile(t0 * BC);
  }

  for (int j = t0; j < nt; ++j) {
    const int k0 = j * BC;

    // ================= S^T = K.Q^T over this wave's part of D =================
    // LDS reads run PF1 MFMAs ahead of their consumer; sched_barrier fences pin that
    // order (left alone, the scheduler hoists every read and spills).
    f32x16 sacc[NKB];
    {
      constexpr int N1 = KS * NKB;
      v8 kf[N1];
      auto k_frag = [&](int n) -> v8 {
        const int s = n / NKB, kb = n % NKB;  // d-step outer: consecutive MFMAs alternate the S accumulators
        return *(FFPA_LDS const v8*)(kaddr[s & 7] + (s >> 3) * 256 + kb * 32 * RB);
      };
      const int dlane = opaque_lane(lane);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < PF1 && n < N1; ++n) kf[n] = k_frag(n);
      static_for<N1>([&](auto ic) {
        constexpr int n = decltype(ic)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (n + PF1 < N1) kf[n + PF1] = k_frag(n + PF1);
        if constexpr (kInterleave && n % kStep == 0 && n / kStep < PPW) {
          // V(j) streams in under this tile's QK^T (the V buffer is free since barrier B of tile j-1)
          issue_v(std::integral_constant<int, n / kStep>{}, k0, dlane);
        }
        constexpr int s = n / NKB, kb = n % NKB;
        if constexpr (s == 0) E::template mfma_v_first<kPad>(sacc[kb], kf[n], qf[s]);
        else E::template mfma_v_acc<kPad>(sacc[kb], kf[n], qf[s]);
      });
      // MFMA result -> VALU reader wait states (invisible to the compiler inside asm)
      if constexpr (NKB == 2) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(sacc[0]), "+v"(sacc[1]));
      else asm volatile("s_nop 15\n\ts_nop 3" : "+v"(sacc[0]));
      __builtin_amdgcn_sched_barrier(0);
    }

    // 16-bit bias tile of this step (two 16-byte loads per lane and key block): issued here, ahead of barrier A and of
    // the K(j+1) burst, so that their latency overlaps both; consumed in the score-modifier section below.
    constexpr bool kBiasEarly = !SAFE && ND > 1;  // measured: -10 % at D = 1024; D <= 512 has no registers to spare (+2 %)
    u32x4 braw[kBiasEarly ? 2 * NKB : 1];
    const bool mask_free = k0 >= free_lo && k0 + BC <= free_hi;  // wave-uniform: this tile lies in the mask's neutral interior
    const bool bias_early = MASK && !kBoolOnly && kBiasEarly && a.bias_vec == 8 && a.bias_dtype != 4 && a.bias_lds == 0 && a.bias_tile == 0 && k0 + BC <= a.Nkv && !mask_free;
    if constexpr (kBiasEarly) {
      if (bias_early) {
        const char* bp = (const char*)a.bias + 2 * (b * a.sbias[0] + hq * a.sbias[1] + (int64_t)qrow_c * a.sbias[2] + k0 + 16 * h);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
          braw[2 * kb] = *(const u32x4*)(bp + kb * 64);
          braw[2 * kb + 1] = *(const u32x4*)(bp + kb * 64 + 16);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    // The kPre early K(j+1) pieces go out in four groups between the softmax stages (the texture-address
    // unit idles there) instead of as one burst of 4 x kPre pieces per CU right behind barrier A1
    constexpr bool kPreSpread = kSpreadReq && kPre >= 4;
    auto pre_k_group = [&](auto gc) __attribute__((always_inline)) {
      if constexpr (kPreSpread) {
        constexpr int g = decltype(gc)::value;
        const int plane = opaque_lane(lane);
        __builtin_amdgcn_sched_barrier(0);
        static_for<kPre / 4>([&](auto ic) { issue_k(std::integral_constant<int, g * (kPre / 4) + decltype(ic)::value>{}, k0 + BC, plane); });
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if constexpr (ND > 1) {  // publish this wave's partial S^T (lane-linear, conflict free)
      FFPA_LDS char* xw = Xb + wave * (NKB * 4096) + lane * 16;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          f32x4 t = {sacc[kb][4 * r4], sacc[kb][4 * r4 + 1], sacc[kb][4 * r4 + 2], sacc[kb][4 * r4 + 3]};
          *(FFPA_LDS f32x4*)(xw + kb * 4096 + r4 * 1024) = t;
        }
    }

    // barrier A: every wave is done reading K(j); V(j) has landed; partials visible
    if constexpr (kPre > 0) {
      // split form: A1 only frees the K buffer, so the first K(j+1) pieces stream under the softmax;
      // V(j) (issued before them) is awaited by count right before the PV loop (barrier A2 below).
      __syncthreads();
      if constexpr (!kPreSpread) {
        const int plane = opaque_lane(lane);
        static_for<kPre>([&](auto ic) { issue_k(ic, k0 + BC, plane); });
      }
    } else {
      dma_wait_all();
      __syncthreads();
    }
    if constexpr (!kInterleave) {
      if (j + 1 < nt) issue_k_tile(k0 + BC);
    }
    __builtin_amdgcn_sched_barrier(0);
    pre_k_group(std::integral_constant<int, 0>{});

    // V^T fragments for the PV loop below.  V(j) is visible from barrier A on, so the first PF2
    // fragments are requested here and their LDS latency hides under the softmax VALU work.
    constexpr int N2 = NDB * NKS;
    v8 vf[N2];
    auto v_frag = [&](int n) -> v8 {
      const int db = n % NDB, ks = n / NDB;  // key-step outer: consecutive MFMAs rotate over all accumulators
      if constexpr (!SAFE) {
        const int krow = (kVHi && ks >= 4) ? 32 * ((ks - 4) >> 1) + 8 * (ks & 1) : 32 * (ks >> 1) + 8 * (ks & 1);
        FFPA_LDS const char* vp = ((kVHi && ks >= 4) ? vaddr[4 + (db & 3)] : vaddr[db & 3]) + (db >> 2) * 256 + krow * RB;
        const v4 lo = E::tr_read(vp);
        const v4 hi = E::tr_read(vp + 4 * RB);
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      } else {
        v8 r;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int key = 32 * (ks >> 1) + 16 * h + 8 * (ks & 1) + jj;
          const int byte = (dh * DW + db * 32 + l31) * 2;
          const int off = byte ^ (v_slot_swizzle<D>(key) * 16);
          r[jj] = *(FFPA_LDS const T*)(Vt + key * RB + off);
        }
        return r;
      }
    };
    if constexpr (kPre == 0) {
#pragma unroll
      for (int n = 0; n < PF2 && n < N2; ++n) vf[n] = v_frag(n);
      __builtin_amdgcn_sched_barrier(0);
    }

    // lane holds x[kb][r] = score(row qrow, key k0 + 32 kb + 16 h + r)
    float x[NKB][16];
    if constexpr (ND == 2) {
      FFPA_LDS const char* xr = Xb + (wave ^ 1) * (NKB * 4096) + lane * 16;  // (4 KiB per wave and 32-key block, as the four-wave form)
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const f32x4 t = *(FFPA_LDS const f32x4*)(xr + kb * 4096 + r4 * 1024);
#pragma unroll
          for (int e = 0; e < 4; ++e) x[kb][4 * r4 + e] = (sacc[kb][4 * r4 + e] + t[e]) * a.scale_log2;
        }
    } else if constexpr (ND == 4) {
      // sum the four D-quarter partials in a fixed order so that all four waves of the row block see
      // bit-identical scores (their softmax state must agree: each owns a different slice of O^T)
      FFPA_LDS const char* xr = Xb + (wave & ~3) * (NKB * 4096) + lane * 16;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          f32x4 acc4 = *(FFPA_LDS const f32x4*)(xr + kb * 4096 + r4 * 1024);
#pragma unroll
          for (int w = 1; w < 4; ++w) {
            const f32x4 t = *(FFPA_LDS const f32x4*)(xr + w * (NKB * 4096) + kb * 4096 + r4 * 1024);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc4[e] += t[e];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) x[kb][4 * r4 + e] = acc4[e] * a.scale_log2;
        }
    } else {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[kb][r] = sacc[kb][r] * a.scale_log2;
    }

    pre_k_group(std::integral_constant<int, 1>{});
    // ================= score modifiers (split_d.cuh:506-539) =================
    if (bias_early) {
      typedef __attribute__((ext_vector_type(8))) __bf16 b8;
      typedef __attribute__((ext_vector_type(8))) _Float16 h8;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          if (a.bias_dtype == 2) {
            const b8 t = __builtin_bit_cast(b8, braw[2 * kb + w]);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[kb][8 * w + e] += (float)t[e] * 1.4426950408889634f;
          } else {
            const h8 t = __builtin_bit_cast(h8, braw[2 * kb + w]);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[kb][8 * w + e] += (float)t[e] * 1.4426950408889634f;
          }
        }
    } else if (!MASK) {
    } else if (kBoolOnly) {
      if (!mask_free) {
        const int64_t brow = b * a.sbias[0] + hq * a.sbias[1] + (int64_t)qrow_c * a.sbias[2];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
          const int kbase = k0 + kb * 32 + 16 * h;
          if (a.bias_vec == 16 && k0 + BC <= a.Nkv) apply_bool_block_vec(x[kb], a.bias, brow, kbase);
          else apply_bool_block(x[kb], a.bias, brow, a.sbias[3], kbase, a.Nkv);
        }
      }
    } else if (kBiasTile && !mask_free) {
      // bias tile staged by this wave during the previous step's PV loop (drained at barrier B)
      typedef __attribute__((ext_vector_type(8))) __bf16 b8;
      typedef __attribute__((ext_vector_type(8))) _Float16 h8;
      FFPA_LDS const char* brow_lds = Bt + l31 * (BC * 2);
      const int g = bias_swz(l31);
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const u32x4 raw = *(FFPA_LDS const u32x4*)(brow_lds + (((kb * 4 + 2 * h + w) ^ g) << 4));
          if (a.bias_dtype == 2) {
            const b8 t = __builtin_bit_cast(b8, raw);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[kb][8 * w + e] += (float)t[e] * 1.4426950408889634f;
          } else {
            const h8 t = __builtin_bit_cast(h8, raw);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[kb][8 * w + e] += (float)t[e] * 1.4426950408889634f;
          }
        }
    } else if (a.bias_lds > 0 && !mask_free) {
      // key bias from the LDS row cache: every lane of a half reads the same 16 keys (LDS broadcast)
      typedef __attribute__((ext_vector_type(8))) __bf16 b8;
      typedef __attribute__((ext_vector_type(8))) _Float16 h8;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const int key = k0 + kb * 32 + 16 * h;
        if (a.bias_dtype == 3) {
          FFPA_LDS const char* bp = Bl + key * 4;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const f32x4 t = *(FFPA_LDS const f32x4*)(bp + 16 * i);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[kb][4 * i + e] += t[e] * 1.4426950408889634f;
          }
        } else {
          FFPA_LDS const char* bp = Bl + key * 2;
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            const u32x4 raw = *(FFPA_LDS const u32x4*)(bp + 16 * w);
            if (a.bias_dtype == 2) {
              const b8 t = __builtin_bit_cast(b8, raw);
#pragma unroll
              for (int e = 0; e < 8; ++e) x[kb][8 * w + e] += (float)t[e] * 1.4426950408889634f;
            } else {
              const h8 t = __builtin_bit_cast(h8, raw);
#pragma unroll
              for (int e = 0; e < 8; ++e) x[kb][8 * w + e] += (float)t[e] * 1.4426950408889634f;
            }
          }
        }
      }
    } else if (a.bias_dtype != 0 && !mask_free) {
      const int64_t brow = b * a.sbias[0] + hq * a.sbias[1] + (int64_t)qrow_c * a.sbias[2];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const int kbase = k0 + kb * 32 + 16 * h;
        if (a.bias_dtype == 4) {
          if (a.bias_vec == 16 && k0 + BC <= a.Nkv) apply_bool_block_vec(x[kb], a.bias, brow, kbase);
          else apply_bool_block(x[kb], a.bias, brow, a.sbias[3], kbase, a.Nkv);
          continue;
        }
        if (a.bias_vec && k0 + BC <= a.Nkv) {  // full tile, unit key stride, rows aligned to the vector width
          if (a.bias_dtype == 3) add_bias_block_vec<float, 4>(x[kb], a.bias, brow, kbase);
          else if (a.bias_vec == 8) {
            if (a.bias_dtype == 2) add_bias_block_vec<__bf16, 8>(x[kb], a.bias, brow, kbase);
            else add_bias_block_vec<_Float16, 8>(x[kb], a.bias, brow, kbase);
          } else {
            if (a.bias_dtype == 2) add_bias_block_vec<__bf16, 4>(x[kb], a.bias, brow, kbase);
            else add_bias_block_vec<_Float16, 4>(x[kb], a.bias, brow, kbase);
          }
          continue;
        }
        if (a.bias_dtype == 3) add_bias_block<float>(x[kb], a.bias, brow, a.sbias[3], kbase, a.Nkv);
        else if (a.bias_dtype == 2) add_bias_block<__bf16>(x[kb], a.bias, brow, a.sbias[3], kbase, a.Nkv);
        else add_bias_block<_Float16>(x[kb], a.bias, brow, a.sbias[3], kbase, a.Nkv);
      }
    }
    const bool tail = k0 + BC > a.Nkv;
    const bool diag = a.causal && ((int64_t)k0 + BC - 1 > (int64_t)(a.causal_row_mod ? 0 : wq0) + a.causal_offset);
    if (tail || diag) {
      const int crow = a.causal_row_mod ? qrow % a.causal_row_mod : qrow;
      const int64_t lim = a.causal ? (int64_t)crow + a.causal_offset : (int64_t)a.Nkv;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + kb * 32 + 16 * h + r;
          if (key >= a.Nkv || key > lim) x[kb][r] = -INFINITY;
        }
    }

    // ================= online softmax (prefill.cuh:671-870, log2 domain) =================
    float tmax = x[0][0];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, x[kb][r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    pre_k_group(std::integral_constant<int, 2>{});
    const float m_new = fmaxf(m_run, tmax);
    // lazy rescale: keep the stale max while it grew by <= thr (prefill.cuh:684-755);
    // the first finite max always takes the branch (m_run = -inf).
    const bool grow = m_new > m_run + a.thr;
    if (__any(grow)) {
      const float alpha = grow ? __builtin_amdgcn_exp2f(m_run - m_new) : 1.f;
      if (j > t0) {
        // Rare path.  O^T lives in AGPRs, which the VALU cannot address: scale it in place
        // through one temporary VGPR.  Written as asm on "+a" operands so the accumulator
        // never acquires a VGPR live range (which makes hipcc spill the whole hot loop).
#pragma unroll
        for (int i = 0; i < NDB; ++i) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float e = oacc[i][r];
            float t;
            asm volatile(
                "v_accvgpr_read_b32 %1, %0\n\ts_nop 1\n\tv_mul_f32 %1, %1, %2\n\ts_nop 1\n\t"
                "v_accvgpr_write_b32 %0, %1"
                : "+a"(e), "=&v"(t)
                : "v"(alpha));
            oacc[i][r] = e;
          }
        }
      }
      l_run *= alpha;
      m_run = grow ? m_new : m_run;
    }
    float m_use = (m_run == -INFINITY) ? 0.f : m_run;

    v8 pf[NKS];
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(x[kb][r] - m_use);
        psum += p;  // row sum from the unrounded P (prefill.cuh:755-756)
        // contraction slot (h, r & 7) of key step 2 kb + (r >> 3) <-> this register (key 32 kb + 16 h + r)
        pf[kb * 2 + (r >> 3)][r & 7] = (T)p;
      }
    l_run += psum;
    pre_k_group(std::integral_constant<int, 3>{});
    if constexpr (DROP) {
      // applied to the ROUNDED P, after the row sum (LSE is undropped): prefill.cuh:508-546
      const unsigned long long erow =
          a.philox_offset + (((unsigned long long)b * a.Hq + hq) * a.Nq + (unsigned long long)qrow_c) * (unsigned long long)a.Nkv;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_barrier(0);  // one Philox group at a time: bounded register pressure
          float keep[4];
          dropout_keep4(a.philox_seed, erow + (unsigned long long)(k0 + kb * 32 + 16 * h + 4 * i), a.keep_threshold, a.keep_scale, keep);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int r = 4 * i + t;
            const float dropped = (float)pf[kb * 2 + (r >> 3)][r & 7] * keep[t];
            pf[kb * 2 + (r >> 3)][r & 7] = (T)dropped;
          }
        }
    }

    // ================= O^T += V^T.P^T =================
    {
      const int dlane = opaque_lane(lane);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kPre > 0) {
        // barrier A2: V(j) has landed on every wave (all but the kPre younger K pieces have retired)
        dma_wait_except<kPre>();
        __syncthreads();
      }
      if constexpr (kPre > 0) {
#pragma unroll
        for (int n = 0; n < PF2 && n < N2; ++n) vf[n] = v_frag(n);
      }
      static_for<N2>([&](auto ic) {
        constexpr int n = decltype(ic)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (n + PF2 < N2) vf[n + PF2] = v_frag(n + PF2);
        if constexpr (kInterleave && n % kStep == 0 && n / kStep + kPre < PPW) {
          // K(j+1) streams in under this tile's PV (the K buffer is free since barrier A).  After the
          // last tile that is an empty tile (every lane out of range: zeros, no memory traffic): cheaper
          // than a branch per piece, and barrier B still drains it before the workgroup can exit.
          issue_k(std::integral_constant<int, n / kStep + kPre>{}, k0 + BC, dlane);
        }
        if constexpr (kBiasTile && kInterleave) {
          // the bias tile of step j+1 (this wave's rows only, into its private area: no other wave reads or writes it, so the
          // wave's own queue drain at barrier B is all the synchronisation it needs); slots the K pieces leave free
          constexpr int kBStep = kStep >= 2 ? kStep : 2;
          if constexpr (n % kBStep == kBStep / 2 && n / kBStep < kBtPieces) {
            issue_bias(std::integral_constant<int, n / kBStep>{}, k0 + BC);
          }
        }
        constexpr int db = n % NDB, ks = n / NDB;
        oacc[db] = E::mfma(vf[n], pf[ks], oacc[db]);
      });
      if constexpr (kBiasTile && kInterleave) {
        // (short PV loops — small head dims — do not have a slot for every bias piece: the rest go out here)
        constexpr int kBStep = kStep >= 2 ? kStep : 2;
        static_for<kBtPieces>([&](auto ic) {
          if constexpr (decltype(ic)::value * kBStep + kBStep / 2 >= N2) issue_bias(ic, k0 + BC);
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    // barrier B: every wave is done reading V(j); K(j+1) has landed and is visible
    dma_wait_all();
    __syncthreads();
    if constexpr (!kInterleave) {
      if (j + 1 < nt) issue_v_tile(k0 + BC);
    }
  }

  // ================= epilogue (prefill.cuh:1018-1093) =================
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = __builtin_amdgcn_rcpf(l_tot);  // fully masked row: 0 * inf = NaN, as SDPA
  if (qrow < a.Nq) {
    if (a.nsplit > 1) {
      // split-KV partial: normalised fp32 O and its LSE; an empty / fully masked share contributes
      // nothing (O = 0, LSE = -inf).  Merged by ffpa_fwd_merge_kernel.
      const bool dead = !(l_tot > 0.f);
      const int64_t prow = (((int64_t)split * a.B + b) * a.Hq + hq) * a.Nq + qrow;
      float* wp = a.ws_o + prow * D + dh * DW + 4 * h;
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f32x4 w;
#pragma unroll
          for (int t = 0; t < 4; ++t) w[t] = dead ? 0.f : oacc[db][4 * i + t] * inv;
          if (a.tickets != nullptr) store_write_through(wp + db * 32 + 8 * i, w);  // (merged in this launch: see split_arrive_and_merge)
          else *(f32x4*)(wp + db * 32 + 8 * i) = w;
        }
      if (h == 0 && dh == 0) {
        const float lse_part = dead ? -INFINITY : __builtin_fmaf(m_run, 0.6931471805599453f, __logf(l_tot));  // (explicit: see ffpa_fwd_m16_kernel.h)
        if (a.tickets != nullptr) __hip_atomic_store(&a.ws_lse[prow], lse_part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else a.ws_lse[prow] = lse_part;
      }
    } else {
      // The lane owns 4-element groups d = 32 db + 8 i + 4 h + (0..3); its partner lane ^ 32 owns the other half
      // of each 8-element run.  The two trade groups (v_permlane32_swap: this lane's odd group for the partner's
      // even one) so that each stores whole 16-byte runs: the epilogue is store-issue bound, and this halves the
      // number of store instructions.
      T* op = (T*)a.o + b * a.so[0] + hq * a.so[1] + (int64_t)qrow * a.so[2] + dh * DW + 8 * h;
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          v4 even, odd;  // groups i = 2 pr and i = 2 pr + 1
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            even[t] = (T)(oacc[db][8 * pr + t] * inv);
            odd[t] = (T)(oacc[db][8 * pr + 4 + t] * inv);
          }
          const u32x2 e2 = __builtin_bit_cast(u32x2, even), o2 = __builtin_bit_cast(u32x2, odd);
          u32x4 run;  // h = 0: [own even | partner's even]   h = 1: [partner's odd | own odd]
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            const auto sw = __builtin_amdgcn_permlane32_swap(e2[w], o2[w], false, false);
            run[w] = sw[0];
            run[2 + w] = sw[1];
          }
          if (dh * DW + db * 32 + pr * 16 + 8 * h < a.d_valid) *(u32x4*)(op + db * 32 + pr * 16) = run;
        }
      if (a.lse != nullptr && h == 0 && dh == 0) {
        // natural-log LSE = ln(l) + m*ln2 (prefill.cuh:1063-1073)
        a.lse[((int64_t)b * a.Hq + hq) * a.Nq + qrow] = __builtin_fmaf(m_run, 0.6931471805599453f, __logf(l_tot));
      }
    }
  }
  // KV-split launch with tickets: the last split of this row tile to arrive merges all partials here (one launch per call)
  if (a.nsplit > 1 && a.tickets != nullptr) split_arrive_and_merge<T, true>(a, D, bh * a.nqt + qt, b, hq, q0, BR, Kt);
}

// Merge the split-KV partials of one launch (the reference's decode stage 2,
// csrc/cuffpa/native/sm_80/split_kv.cuh:329-455): O = sum_s w_s O_s / sum_s w_s with
// w_s = exp(LSE_s - max_s LSE_s), LSE = max + ln(sum_s w_s).  One 64-lane workgroup per (row, 256-column chunk):
// the lanes first share out the splits (max, weights -> LDS, sum: wave reductions), then each accumulates its 4
// columns over the splits with the loads of 8 splits in flight at a time.  (The first version walked the splits
// serially per lane and recomputed the weights per chunk: 60 us for a 64-split launch, 2.6x the main kernel.)
template <typename T>
__global__ __launch_bounds__(64) void ffpa_fwd_merge_kernel(const FwdArgs a, int D) {
  __shared__ float wsh[kMergeMaxSplits];
  const int64_t row = blockIdx.x;  // (b * Hq + hq) * Nq + qrow
  const int64_t rows = (int64_t)a.B * a.Hq * a.Nq;
  const int lane = threadIdx.x;
  float mx = -INFINITY;
  for (int s = lane; s < a.nsplit; s += 64) mx = fmaxf(mx, a.ws_lse[s * rows + row]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float wsum = 0.f;
  for (int s = lane; s < a.nsplit; s += 64) {
    const float w = (mx == -INFINITY) ? 0.f : __expf(a.ws_lse[s * rows + row] - mx);
    wsh[s] = w;
    wsum += w;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wsum += __shfl_xor(wsum, o);
  __syncthreads();
  const float inv = 1.f / wsum;  // every share empty -> 0 * inf = NaN, like an unsplit fully masked row
  const int d = blockIdx.y * 256 + lane * 4;
  if (d < a.d_valid) {  // D = the kernel's (64-multiple) head dim = the partials' row length; only the caller's columns are stored
    const float* src = a.ws_o + row * D + d;
    const int64_t sstride = rows * D;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 8 <= a.nsplit; s += 8) {
      f32x4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = *(const f32x4*)(src + (s + u) * sstride);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float w = wsh[s + u];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += w * t[u][e];
      }
    }
    for (; s < a.nsplit; ++s) {
      const f32x4 t = *(const f32x4*)(src + s * sstride);
      const float w = wsh[s];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += w * t[e];
    }
    const int qrow = (int)(row % a.Nq);
    const int64_t bh = row / a.Nq;
    const int hq = (int)(bh % a.Hq);
    const int b = (int)(bh / a.Hq);
    T* op = (T*)a.o + b * a.so[0] + hq * a.so[1] + (int64_t)qrow * a.so[2];
    typename Elem<T>::v4 w4;
#pragma unroll
    for (int e = 0; e < 4; ++e) w4[e] = (T)(acc[e] * inv);
    *(typename Elem<T>::v4*)(op + d) = w4;
  }
  if (a.lse != nullptr && lane == 0 && blockIdx.y == 0) a.lse[row] = (mx == -INFINITY) ? -INFINITY : mx + __logf(wsum);
}

}  // namespace ffpa

#include "ffpa_mask_bounds.h"
