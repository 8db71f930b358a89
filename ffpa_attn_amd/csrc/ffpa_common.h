// ffpa_common.h — what every kernel of this library shares: vector types, the compile-time loop, the launch arguments (FwdArgs), the per-dtype MFMA /
// transpose-read wrappers (Elem), the LDS swizzles, the LDS-DMA forms (buffer_load ... lds) with their counted waits, buffer descriptors and tile sources.
// Split out of ffpa_fwd_kernel.h at the end of round 5 (text moved, nothing changed: every object rebuilds byte-identical — profiles/r05_fold_manifest.txt).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#define FFPA_M0_CLOBBER , "m0"  // the LDS-DMA asm writes M0 behind the compiler's back and says so

// The S^T MFMAs are inline asm, so hipcc's hazard recognizer does not see their operands (a VALU write needs 2
// wait states before an MFMA reads the register).  In the product kernels their A operand comes from ds_read
// (ordered by lgkmcnt), B (the Q fragments) is written once per workgroup by global loads, and C is the previous
// MFMA of the chain: no VALU write precedes any of them, so nothing is padded in (a `s_nop 1` per MFMA cost 1.4 %
// at D = 512, 2.7 % at D = 320).  tools/check_mfma_hazards.py proves the "no VALU write within the last two
// instructions" property on the generated ISA of every instantiation.  The register-staged SAFE twins (tests
// only) do get the pad: there hipcc parks Q fragments in spare AGPRs and restores them right before the MFMA.
#define FFPA_MFMA_PAD "s_nop 1\n\t"

namespace ffpa {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{}) — the
// index is a constant expression inside the body (`if constexpr` on it, static register-array indices).
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

#define FFPA_LDS __attribute__((address_space(3)))
#define FFPA_GLB __attribute__((address_space(1)))

constexpr unsigned kFlagNoXcdRemap = 0x2u;
constexpr unsigned kFlagStreamKV = 0x80000000u;  // set by the launch side only (ffpa_capi.hip): every K / V byte of this launch is read by ONE workgroup and the
                                                 // K + V of the launch exceed the Infinity Cache — the short-query tiles then fetch them with the non-temporal hint

// Keys per tile of the tiles whose head dim is split over waves (ND > 1): 32 — except the short-query tiles (ND = 4) of head dims 128, 384 and 512,
// which take 64: they are HBM-bound and what they stream per request burst is one tile (D = 512: 32 KiB at 32 keys reached 5.5 TB/s where the
// 64 KiB tiles of D = 1024 reach 6.1).  D = 384 / 512 run ONE workgroup per CU (the split rule for these head dims, ffpa_capi.hip), so the LDS has
// the room; D = 128 still fits two.  D = 256 does not (96 KiB per workgroup: one per CU, + 6 % time: measured, profiles/r03_decode_splits.txt).
// The partial-S exchange area grows with the tile: 4 KiB per wave and 32-key block.
// Round 4: the two-wave split (ND = 2: head dims that are not multiples of 128) takes 64-key tiles too where the LDS has the room — D = 320 (112 KiB)
// and D = 448 (144 KiB); 576 and up would need 176 KiB.  Interleaved A/B on one box (profiles/r04_decode_nd2.txt): D = 320 B1 H32 Nkv 8192 75.9 -> 66.9 us
// (4.4 -> 5.0 TB/s of K / V), B8 GQA 144 -> 123 us (5.45 TB/s), 64k keys 144 -> 124 us, 16 query rows - 11 %; D = 448 - 2 % / +- 0.
constexpr int splitd_block_keys(int D, int ND) {
  return ((ND == 4 && (D == 128 || D == 384 || D == 512)) || (ND == 2 && (D == 320 || D == 448))) ? 64 : 32;
}
constexpr int splitd_exchange_bytes(int D, int ND) { return ND > 1 ? 4 * 4096 * (splitd_block_keys(D, ND) / 32) : 0; }

// Workgroup id -> position in the launch's logical order (batch-major, head, row tile, split).  The hardware deals workgroup ids round-robin
// to the 8 XCDs (id & 7); each XCD has its own L2.  `group` (FwdArgs::xcd_group: 1, 2, 4 or 8) XCDs share a contiguous range of the logical
// order and take its workgroups in turn:
//   group 1: an XCD walks a contiguous range — all row tiles of a head stream K/V through ONE L2 (the default);
//   group 8: the logical order is the id order — every head is spread over all eight XCDs;
//   in between: `group` XCDs work on the same head, 8 / group heads are in flight chip-wide.
// Why it is a launch-side choice: 8 / group heads' K + V are what the 256 MiB Infinity Cache has to hold for the second and later rounds of
// row tiles to be served from it instead of HBM (ffpa_capi.hip picks the smallest group that fits).  Handles totals that are not multiples of 8.
__device__ __forceinline__ int xcd_logical_id(int id, int total, int group) {
  const int xcd = id & 7, j = id >> 3, per = total >> 3, rem = total & 7;
  const int first = xcd & ~(group - 1);                       // first XCD of this XCD's group
  const int gstart = first * per + (first < rem ? first : rem);  // ids owned by the XCDs in front of the group
  return gstart + j * group + (xcd - first);
}

// Kernel argument block (host fills it from ffpa_fwd_params).
struct FwdArgs {
  const void* q;
  const void* k;
  const void* v;
  void* o;
  float* lse;
  const void* bias;
  int64_t sq[3], sk[3], sv[3], so[3];  // element strides: batch, head, row
  int64_t sbias[4];                    // element strides: batch, head, row, key
  int B, Hq, Hkv, Nq, Nkv;
  int total_wg;       // B * Hq * nqt * nsplit: the (batch, head, row tile, split) ids of the launch (== gridDim.x unless persistent)
  int d_valid;        // the caller's head dim (a multiple of 8, <= the kernel's D): columns [d_valid, D) read as zeros and are never stored
  int group;          // Hq / Hkv
  int nqt;            // row tiles per (batch, head)
  int bias_dtype;     // 0 none, 1 fp16, 2 bf16, 3 fp32 (additive), 4 bool8 (byte != 0 <=> key visible, else -inf)
  int causal;
  int causal_offset;  // visible iff key <= row + causal_offset
  float scale_log2;   // softmax_scale * log2(e); the 16x16x32 build only ever sees a POSITIVE value here (q_mode below)
  float inv_scale;    // 16x16x32 build: 1 / (the scale behind scale_log2) — additive biases enter the S^T accumulators in units of 1 / scale
  int q_mode;         // 16x16x32 build: 0 = Q as stored; 1 = Q fragments zeroed (softmax_scale == 0: scores = 0 * q.k + bias, scale pair (log2 e, 1));
                      //   2 = Q fragments negated (softmax_scale < 0: scores = |scale| * (-q).k + bias, scale pair (|scale| log2 e, 1 / |scale|))
  float thr;          // lazy-rescale threshold, log2 units (0 = exact recurrence)
  unsigned flags;
  // split-KV (short-query / decode launches): workgroup (tile, split) handles KV tiles
  // [split * tiles_per_split, ...) and writes a normalised fp32 partial + its LSE to the workspace
  int nsplit;           // 1 = no split (write O / LSE directly)
  int tiles_per_split;
  float* ws_o;          // [nsplit, B, Hq, Nq, D] fp32
  float* ws_lse;        // [nsplit, B, Hq, Nq]    fp32
  int* tickets;         // [B * Hq * nqt] zeroed counters, or NULL: with them the last split of a row tile to arrive merges the partials in this launch
  // rows of several query heads of one KV group packed into one row axis (host reshape): the causal
  // limit of packed row r is (r % causal_row_mod) + causal_offset; 0 = rows are plain query rows
  int causal_row_mod;
  // optional [first, end) visible-key bounds per block of 32 query rows (see ffpa_fwd_params.kv_bounds)
  const int* kv_bounds;
  int64_t s_bounds[2];  // element strides: batch, head (0 = broadcast)
  int bias_lds;  // < 0: -bytes of LDS reserved for bias_tile staging.  > 0: the bias is a key bias (no row axis): its [Nkv] row of this (batch, head) is copied to LDS once per workgroup
                 //      (this many bytes, a whole number of tiles) and the tiles read it from there instead of from global memory
  int bias_cache_raw;  // 16x16x32 build, bias_lds > 0: 1 = the row cache holds the caller's 16-bit elements (a key row too long for the fp32 / scale form:
                       //   converted at the top of every KV step), 0 = fp32 values already divided by the scale
  int bias_tile; // 1: a 16-bit bias with a real row axis is staged through LDS: every wave LDS-DMAs the [32 rows x BC keys] tile of
                 //    the NEXT step into a private area while the PV MFMAs run, and reads it there when it is needed
  int bias_vec;  // W in {0, 4, 8, 16}: bias key stride is 1 and base / strides are W-element aligned -> W-wide loads (16: bool8 masks)
  // dropout (prefill.cuh:398-546): keep iff u > p, u from Philox4x32-10 at the logical element offset
  float dropout_p;          // 0 = off
  float keep_scale;         // 1 / (1 - p)
  int xcd_group;            // XCDs that share a contiguous range of the launch's logical workgroup order: 1 (default), 2, 4 or 8 (xcd_logical_id)
  int l2_prefetch;          // 16x16x32 prefill builds: touch the K/V tile two steps ahead (ffpa_fwd_m16_kernel.h; ffpa_capi.hip decides)
  uint32_t keep_threshold;  // smallest Philox word whose element is kept: word >= keep_threshold <=> ((float)word + 1.0f) * 2^-32 > dropout_p
  unsigned long long philox_seed;
  unsigned long long philox_offset;
  int pair_tiles;           // 16x16x32 prefill builds under the causal flag: a workgroup walks row tile nqt - 1 - i and then row tile i (the grid holds (nqt + 1) / 2 workgroups per
                            // head).  LAST on purpose: the field offsets in front of it — and with them every kernel that does not read it — stay what they were.
};

template <typename T>
struct Elem;

template <>
struct Elem<__bf16> {
  typedef __attribute__((ext_vector_type(8))) __bf16 v8;
  typedef __attribute__((ext_vector_type(4))) __bf16 v4;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  // VGPR-form MFMA for the S^T accumulator (hipcc selects the AGPR form for every builtin
  // MFMA of a kernel, and the 256 AGPRs are exactly the O^T accumulator).  "s_nop 1" covers
  // the VALU-write -> MFMA-operand wait states the compiler cannot see inside asm.
  template <bool PAD>
  static __device__ __forceinline__ void mfma_v_first(f32x16& d, v8 a, v8 b) {
    if constexpr (PAD) asm volatile(FFPA_MFMA_PAD "v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
  }
  template <bool PAD>
  static __device__ __forceinline__ void mfma_v_acc(f32x16& d, v8 a, v8 b) {
    if constexpr (PAD) asm volatile(FFPA_MFMA_PAD "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
  }
  static __device__ __forceinline__ v4 tr_read(FFPA_LDS const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((FFPA_LDS v4*)p);
  }
};

template <>
struct Elem<_Float16> {
  typedef __attribute__((ext_vector_type(8))) _Float16 v8;
  typedef __attribute__((ext_vector_type(4))) _Float16 v4;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  template <bool PAD>
  static __device__ __forceinline__ void mfma_v_first(f32x16& d, v8 a, v8 b) {
    if constexpr (PAD) asm volatile(FFPA_MFMA_PAD "v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
  }
  template <bool PAD>
  static __device__ __forceinline__ void mfma_v_acc(f32x16& d, v8 a, v8 b) {
    if constexpr (PAD) asm volatile(FFPA_MFMA_PAD "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
  }
  static __device__ __forceinline__ v4 tr_read(FFPA_LDS const char* p) {
    typedef __attribute__((ext_vector_type(4))) short s4;
    const s4 raw = __builtin_amdgcn_ds_read_tr16_b64_v4i16((FFPA_LDS s4*)p);
    return __builtin_bit_cast(v4, raw);
  }
};

// ---------------------------------------------------------------------------------
// LDS image of a K / V tile: row-major [BC][D] with a per-row XOR swizzle of the
// 16-byte slot index (K) / the 64-byte quarter (V).  Row stride D*2 bytes.
//   K is read with ds_read_b128 by 16-lane groups whose rows are distinct mod 16 and
//   whose column slot is equal -> XOR the slot with a row hash that is a bijection
//   over row mod 16 onto the slot's bank position.
//   V is read with ds_read_b64_tr_b16: a 32-lane half reads 4 keys x 64 bytes -> the
//   4 keys must land in the 4 different 64-byte quarters of the 256-byte bank row.
// ---------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ int k_slot_swizzle(int key) {
  return (D % 128 == 0) ? (key & 15) : ((key >> 1) & 7);
}
template <int D>
__device__ __forceinline__ int v_slot_swizzle(int key) {  // in 16-byte slots
  return (D % 128 == 0) ? ((key & 3) << 2) : (((key >> 1) & 1) << 2);
}

// One 1 KiB LDS-DMA piece: buffer_load_dwordx4 ... lds (16 B per lane, destination M0 + lane*16).
// Issued through inline asm on purpose: hipcc cannot tell that the DMA's LDS write does not alias a
// later ds_read_b64_tr_b16 and would put `s_waitcnt vmcnt(0)` in front of every transpose read that
// follows a builtin DMA (measured: 561 vs 1030 TFLOP/s).  The asm is invisible to that pass, so the
// kernel drains the DMA queue itself (dma_wait_all) before each workgroup barrier.  Compiler-counted
// vmcnt waits for its own loads stay correct (loads retire in order; hidden younger ops only make a
// counted wait conservative).  M0 is written in the same statement that consumes it and is declared clobbered (the compiler
// keeps no value of its own in M0 across the statement; tools/check_mfma_hazards.py still verifies that nothing else writes it).
// NT: the non-temporal hint (`nt`): a stream that nobody reads twice — the K / V of a short-query launch — goes through L2 / MALL without displacing
// anything and without the fill traffic of a line that is never hit: LDS-DMA streaming from HBM reaches 5.9 TB/s without the hint and 7.3 TB/s
// with it (tools/probes/hbm_read_probe.hip, profiles/r04_hbm_read_probe.txt).  Prefill tiles are re-read by the other row tiles of the head: no hint.
template <bool NT = false>
__device__ __forceinline__ void lds_dma_16(u32x4 rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  if constexpr (NT) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory" FFPA_M0_CLOBBER);
  } else {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory" FFPA_M0_CLOBBER);
  }
}

// NT_BUILD builds pick the form per launch (`stream`, wave-uniform: one scalar branch per piece in a kernel that waits on HBM).
template <bool NT_BUILD>
__device__ __forceinline__ void lds_dma_16_sel(bool stream, u32x4 rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  if constexpr (NT_BUILD) {
    if (stream) {
      lds_dma_16<true>(rsrc, lds_addr, voff, soff);
      return;
    }
  }
  lds_dma_16<false>(rsrc, lds_addr, voff, soff);
}

// The same with the destination given as scalar base + compile-time constant (nothing to precompute and keep in a register per piece).
template <int LCONST>
__device__ __forceinline__ void lds_dma_16_at(u32x4 rsrc, uint32_t lds_base, uint32_t voff, uint32_t soff) {
  asm volatile("s_add_u32 m0, %0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff), "n"(LCONST)
               : "memory", "scc" FFPA_M0_CLOBBER);
}

// ... and with the non-temporal hint, chosen at compile time: what the packed-sequence kernel's decode-batch build issues (ffpa_fwd_m16_kernel.h: the tile text names
// its DMA form through a macro of the enclosing kernel; the dense kernels name lds_dma_16_at itself)
template <bool NT>
struct LdsDma16 {
  template <int LCONST>
  static __device__ __forceinline__ void at(u32x4 rsrc, uint32_t lds_base, uint32_t voff, uint32_t soff) {
    if constexpr (NT) {
      asm volatile("s_add_u32 m0, %0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds"
                   :
                   : "s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff), "n"(LCONST)
                   : "memory", "scc" FFPA_M0_CLOBBER);
    } else {
      lds_dma_16_at<LCONST>(rsrc, lds_base, voff, soff);
    }
  }
};

// Row-uniform form (one LDS image row == whole pieces: D = 512).  Everything but the per-lane swizzled
// column offset `voff` is scalar: destination = lds_base + LCONST (+ IMM), source row offset `row_off` inside
// the tile, and IMM advances source and destination together for the second KiB of a 2 KiB row.  No VALU;
// rows past the tile's last key are zero-filled by the descriptor's range check.
template <int LCONST, int IMM, bool NT = false>
__device__ __forceinline__ void lds_dma_row(u32x4 rsrc, uint32_t lds_base, uint32_t voff, uint32_t row_off) {
  if constexpr (NT) {
    asm volatile(
        "s_add_u32 m0, %0, %4\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen offset:%5 nt lds"
        :
        : "s"(lds_base), "v"(voff), "s"(rsrc), "s"(row_off), "n"(LCONST), "n"(IMM)
        : "memory", "scc" FFPA_M0_CLOBBER);
  } else {
    asm volatile(
        "s_add_u32 m0, %0, %4\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen offset:%5 lds"
        :
        : "s"(lds_base), "v"(voff), "s"(rsrc), "s"(row_off), "n"(LCONST), "n"(IMM)
        : "memory", "scc" FFPA_M0_CLOBBER);
  }
}

// The same with the source row offset computed in place: row_bytes * KEY + base_off (KEY a compile-time row of the tile, base_off the
// wave's first row) — two scalar instructions instead of a table of one scalar register per staged row (BC / 4 for K and as many for V:
// registers the mask / bias builds do not have; spilled, every entry costs a v_readlane + its hazard wait states in front of the piece).
template <int LCONST, int KEY>
__device__ __forceinline__ void lds_dma_row_at(u32x4 rsrc, uint32_t lds_base, uint32_t voff, uint32_t row_bytes, uint32_t base_off) {
  uint32_t tmp;
  asm volatile(
      "s_mul_i32 %0, %4, %6\n\t"
      "s_add_u32 m0, %1, %7\n\t"
      "s_add_u32 %0, %0, %5\n\t"
      "buffer_load_dwordx4 %2, %3, %0 offen lds"
      : "=&s"(tmp)
      : "s"(lds_base), "v"(voff), "s"(rsrc), "s"(row_bytes), "s"(base_off), "n"(KEY), "n"(LCONST)
      : "memory", "scc" FFPA_M0_CLOBBER);
}

// s_waitcnt vmcnt(0) as a BUILTIN (gfx9 encoding 0x0F70: vmcnt = 0, expcnt / lgkmcnt = no wait): the
// compiler's own scoreboard then knows its earlier loads (the Q fragments) have retired and emits no
// counted vmcnt waits inside the tile loop — those would also wait on the hidden DMA pieces.
__device__ __forceinline__ void dma_wait_all() {
  __builtin_amdgcn_s_waitcnt(0x0F70);
  asm volatile("" ::: "memory");
}

// s_waitcnt vmcnt(N): everything but the N youngest VMEM operations has retired (loads retire in order).
// gfx9 encoding: vmcnt[3:0] in bits 3:0, vmcnt[5:4] in bits 15:14.  Builtin form for the same reason as above.
template <int N>
__device__ __forceinline__ void dma_wait_except() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
}

// Raw buffer descriptor over `bytes` bytes at `base` (gfx950: word3 0x00020000 = 32-bit raw dwords).
__device__ __forceinline__ u32x4 make_rsrc(const void* base, uint32_t bytes) {
  const uint64_t a = (uint64_t)base;
  u32x4 r = {(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, bytes, 0x00020000u};
  return r;
}

// Source of one KV tile: a descriptor whose base is the tile's first row, so that every DMA offset is
// tile-relative and 32-bit no matter how large the (batch, head) slice is (a token-major [B,N,H,D] cache
// passes 4 GiB at 128k tokens); only BC rows must span < 4 GiB.  num_records covers exactly the `rows` valid
// rows: gfx950 range-checks voffset + soffset against it and an out-of-range LDS-DMA lane writes ZEROS to LDS
// (tools/probes/lds_dma_oob.hip), i.e. rows past the last key are zero-filled like the reference's cp.async
// staging (prefill.cuh:123-137) — no clamping, no tail branch.  A tile that starts at or past the end (the
// unused prefetch after the last tile) has rows = 0: all zeros, no memory traffic.
struct TileSrc {
  u32x4 rsrc;
  const char* base;
  int rows;
};
template <int BC>
__device__ __forceinline__ TileSrc tile_src(const void* slice, uint32_t row_bytes, int key0, int nkv, uint32_t RB) {
  // min / max only: a select here is lowered to VALU by hipcc and the descriptor then lands in VGPRs
  const int kc = key0 < nkv ? key0 : nkv;
  int rows = nkv - kc;
  rows = rows < BC ? rows : BC;  // 0 when the tile starts at or past the end
  const uint32_t span = (uint32_t)(rows < 1 ? rows : 1) * ((uint32_t)(rows - 1) * row_bytes + (uint32_t)RB);
  TileSrc t;
  t.base = (const char*)slice + (uint64_t)(uint32_t)kc * row_bytes;
  t.rows = rows;
  t.rsrc = make_rsrc(t.base, span);
  return t;
}

// Piece i of this wave's share of one [BC][D] tile whose first row is `base` and which has `rows` valid rows
// (the rest reads as zeros).  Each wave moves TILE/4 bytes as 1 KiB LDS-DMA pieces: lane l of piece p lands at
// lds_tile + p*1024 + l*16 (the hardware's lane-linear rule), so the swizzle goes on the per-lane SOURCE
// offset.  SAFE = the register-staged twin used by the tests.
// kDmaOob: a per-lane offset no tile reaches (tile spans are < 2 GiB and 32-bit offset sums cannot wrap): the range check
// zero-fills that lane.  Used for the K columns at and past the caller's head dim when it is not a multiple of 64 (`slots_valid`
// 16-byte slots per row hold data).  V would not need it — a column of V only ever reaches the same column of O^T, and
// columns past the head dim are not stored — so the hoisted / row-uniform V offsets are left alone (what they fetch there
// is in range: the descriptor ends with the last valid row's last valid byte); the per-piece form masks both.
constexpr uint32_t kDmaOob = 0x80000000u;
template <typename T, int D, int BC, bool IS_V, bool SAFE, bool NT_BUILD = false>
__device__ __forceinline__ void stage_piece(u32x4 rsrc, const char* __restrict__ base, uint32_t row_bytes, int rows,
                                            FFPA_LDS char* lds_tile, int wave, int lane, int i, int slots_valid, bool stream = false) {
  constexpr int SPR = D / 8;  // 16-byte slots per row
  constexpr int PPW = BC * D * 2 / 4096;
  const int p = wave * PPW + i;
  uint32_t voff, soff;
  int key;
  if constexpr ((D * 2) % 1024 == 0) {
    // a row is a whole number of pieces: the row (and its swizzle) is wave-uniform
    constexpr int RPP = D * 2 / 1024;
    key = p / RPP;
    const int sw = IS_V ? v_slot_swizzle<D>(key) : k_slot_swizzle<D>(key);
    voff = (uint32_t)((lane ^ sw) << 4);
    soff = (uint32_t)key * row_bytes + (uint32_t)(p % RPP) * 1024u;
    if ((p % RPP) * 64 + (lane ^ sw) >= slots_valid) voff = kDmaOob;
  } else {
    const int g = p * 64 + lane;
    key = g / SPR;
    const int slot = g - key * SPR;
    const int src_slot = slot ^ (IS_V ? v_slot_swizzle<D>(key) : k_slot_swizzle<D>(key));
    voff = (uint32_t)key * row_bytes + (uint32_t)(src_slot << 4);
    soff = 0;
    if (src_slot >= slots_valid) voff = kDmaOob;
  }
  if constexpr (!SAFE) {
    lds_dma_16_sel<NT_BUILD>(stream, rsrc, (uint32_t)(uintptr_t)(lds_tile + p * 1024), voff, soff);
  } else {
    u32x4 x = {0u, 0u, 0u, 0u};
    if (key < rows && voff != kDmaOob) x = *(const u32x4*)(base + (size_t)voff + (size_t)soff);
    *(FFPA_LDS u32x4*)(lds_tile + p * 1024 + lane * 16) = x;
  }
}

// Keep the per-lane offset arithmetic inside the tile loop: hoisted, it costs dozens of long-lived
// VGPRs that get spilled, and every reload drains the DMA queue (vmcnt(0)).
__device__ __forceinline__ int opaque_lane(int lane) {
  asm volatile("" : "+v"(lane));
  return lane;
}

}  // namespace ffpa
