// ffpa_mask_bounds.h — the mask-range scan (ffpa_attn_mask_kv_bounds): per block of 32 query rows, the keys any row can see and the first run of keys on which the
// mask does nothing — what lets the attention kernels skip hidden KV tiles and the mask reads of neutral ones.
#pragma once

#include "ffpa_common.h"

namespace ffpa {

// Visible-key bounds of a mask (ffpa_fwd_params.kv_bounds): one workgroup per (batch, head, block of 32 query rows) scans
// its 32 x Nkv slab once (coalesced along the keys) and writes four ints:
//   {first, end}            the keys that are visible (not -inf / not False) for at least one row of the block ({Nkv, 0}: none)
//   {free_first, free_end}  the first run of keys for which the mask is NEUTRAL (adds exactly 0 / True) for every row of the block
//                           ({0, 0}: none) — the tiles inside it need no mask at all.
// HBM-bound, 1-4 bytes per mask element.  The per-key "neutral for all 32 rows" flags go through an LDS bitmap (Nkv bits).
template <typename BT>
__device__ __forceinline__ bool mask_elem_visible(BT x);
template <>
__device__ __forceinline__ bool mask_elem_visible<float>(float x) { return __builtin_bit_cast(uint32_t, x) != 0xff800000u; }
template <>
__device__ __forceinline__ bool mask_elem_visible<__bf16>(__bf16 x) { return __builtin_bit_cast(uint16_t, x) != (uint16_t)0xff80u; }
template <>
__device__ __forceinline__ bool mask_elem_visible<_Float16>(_Float16 x) { return __builtin_bit_cast(uint16_t, x) != (uint16_t)0xfc00u; }
template <>
__device__ __forceinline__ bool mask_elem_visible<uint8_t>(uint8_t x) { return x != 0; }

template <typename BT>
__device__ __forceinline__ bool mask_elem_neutral(BT x);  // +0.0 / -0.0 for additive masks, True for boolean ones
template <>
__device__ __forceinline__ bool mask_elem_neutral<float>(float x) { return (__builtin_bit_cast(uint32_t, x) & 0x7fffffffu) == 0u; }
template <>
__device__ __forceinline__ bool mask_elem_neutral<__bf16>(__bf16 x) { return (__builtin_bit_cast(uint16_t, x) & 0x7fffu) == 0u; }
template <>
__device__ __forceinline__ bool mask_elem_neutral<_Float16>(_Float16 x) { return (__builtin_bit_cast(uint16_t, x) & 0x7fffu) == 0u; }
template <>
__device__ __forceinline__ bool mask_elem_neutral<uint8_t>(uint8_t x) { return x != 0; }

struct MaskBoundsArgs {
  const void* bias;
  int64_t sb[4];  // element strides: batch, head, row, key (0 = broadcast)
  int hb, nq, nkv, nblk;
  int words;      // LDS bitmap words = ceil(nkv / 32), 0 = bitmap does not fit (no neutral range is reported)
  int* out;       // [bb, hb, nblk, 4]
};

// Shared tail of both scan kernels: reduce {first, end} over the workgroup, find the first run of set bits in the
// neutral-key bitmap, write the four results.
__device__ __forceinline__ void mask_bounds_finish(const MaskBoundsArgs& m, FFPA_LDS uint32_t* bits, int first, int end) {
  __shared__ int red[4][4];
  const int tid = threadIdx.x, wv = tid >> 6;
  auto wave_min = [](int x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int y = __shfl_xor(x, o);
      x = x < y ? x : y;
    }
    return x;
  };
  auto wave_max = [](int x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int y = __shfl_xor(x, o);
      x = x > y ? x : y;
    }
    return x;
  };
  first = wave_min(first);
  end = wave_max(end);
  __syncthreads();  // the bitmap is complete
  // first neutral key
  int ff = m.nkv;
  for (int w = tid; w < m.words; w += 256) {
    const uint32_t x = bits[w];
    if (x != 0u) {
      const int c = w * 32 + __builtin_ctz(x);
      ff = ff < c ? ff : c;
    }
  }
  ff = wave_min(ff);
  if ((tid & 63) == 0) {
    red[0][wv] = first;
    red[1][wv] = end;
    red[2][wv] = ff;
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    first = first < red[0][w] ? first : red[0][w];
    end = end > red[1][w] ? end : red[1][w];
    ff = ff < red[2][w] ? ff : red[2][w];
  }
  // first key at or after ff that is not neutral
  int fe = m.nkv;
  for (int w = (ff >> 5) + tid; w < m.words; w += 256) {
    uint32_t x = ~bits[w];
    if (w == (ff >> 5)) x &= ~0u << (ff & 31);
    if (x != 0u) {
      const int c = w * 32 + __builtin_ctz(x);
      fe = fe < c ? fe : c;
    }
  }
  fe = wave_min(fe);
  if ((tid & 63) == 0) red[3][wv] = fe;
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int w = 0; w < 4; ++w) fe = fe < red[3][w] ? fe : red[3][w];
    fe = fe < m.nkv ? fe : m.nkv;
    const bool none = m.words == 0 || ff >= m.nkv;
    int* o = m.out + 4 * (int64_t)blockIdx.x;
    o[0] = first;
    o[1] = end;
    o[2] = none ? 0 : ff;
    o[3] = none ? 0 : fe;
  }
}

template <typename BT>
__global__ __launch_bounds__(256) void ffpa_mask_kv_bounds_kernel(const MaskBoundsArgs m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  FFPA_LDS uint32_t* bits = (FFPA_LDS uint32_t*)smem;
  for (int w = threadIdx.x; w < m.words; w += 256) bits[w] = 0u;
  __syncthreads();
  const int blk = blockIdx.x % m.nblk;
  const int bh = blockIdx.x / m.nblk;
  const int h = bh % m.hb, b = bh / m.hb;
  const BT* base = (const BT*)m.bias + b * m.sb[0] + h * m.sb[1];
  const int r0 = blk * 32;
  const int r1 = r0 + 32 < m.nq ? r0 + 32 : m.nq;
  int first = m.nkv, end = 0;
  // all 32 rows of a column are loaded before any is tested: 32 (x2 columns) independent loads in flight per lane
  for (int c0 = threadIdx.x; c0 < m.nkv; c0 += 512) {
    BT x[2][32];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = c0 + 256 * u < m.nkv ? c0 + 256 * u : c0;
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        const int rr = r0 + r < r1 ? r0 + r : r1 - 1;
        x[u][r] = base[(int64_t)rr * m.sb[2] + (int64_t)c * m.sb[3]];
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = c0 + 256 * u;
      bool vis = false, neutral = true;
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        vis = vis || mask_elem_visible<BT>(x[u][r]);
        neutral = neutral && mask_elem_neutral<BT>(x[u][r]);
      }
      if (c < m.nkv) {
        if (vis) {
          first = first < c ? first : c;
          end = end > c + 1 ? end : c + 1;
        }
        if (neutral && m.words) __hip_atomic_fetch_or(&bits[c >> 5], 1u << (c & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  mask_bounds_finish(m, bits, first, end);
}

// 16-byte variant (unit key stride, 16-byte aligned rows, Nkv a multiple of the vector width): each lane owns W = 16
// (bytes), 8 (16-bit) or 4 (fp32) consecutive keys, so a wave reads 1 KiB of a row per load instead of 64-256 B.
template <typename BT>
__global__ __launch_bounds__(256) void ffpa_mask_kv_bounds_vec_kernel(const MaskBoundsArgs m) {
  constexpr int W = 16 / (int)sizeof(BT);
  typedef __attribute__((ext_vector_type(W))) BT bvec;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  FFPA_LDS uint32_t* bits = (FFPA_LDS uint32_t*)smem;
  for (int w = threadIdx.x; w < m.words; w += 256) bits[w] = 0u;
  __syncthreads();
  const int blk = blockIdx.x % m.nblk;
  const int bh = blockIdx.x / m.nblk;
  const int h = bh % m.hb, b = bh / m.hb;
  const BT* base = (const BT*)m.bias + b * m.sb[0] + h * m.sb[1];
  const int r0 = blk * 32;
  const int r1 = r0 + 32 < m.nq ? r0 + 32 : m.nq;
  int first = m.nkv, end = 0;
  for (int c = threadIdx.x * W; c < m.nkv; c += 256 * W) {
    bvec x[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      const int rr = r0 + r < r1 ? r0 + r : r1 - 1;
      x[r] = *(const bvec*)(base + (int64_t)rr * m.sb[2] + c);
    }
    uint32_t nbits = 0u;  // W <= 16 consecutive keys: they sit inside one bitmap word (c is a multiple of W, W divides 32)
#pragma unroll
    for (int e = 0; e < W; ++e) {
      bool vis = false, neutral = true;
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        vis = vis || mask_elem_visible<BT>(x[r][e]);
        neutral = neutral && mask_elem_neutral<BT>(x[r][e]);
      }
      if (vis) {
        first = first < c + e ? first : c + e;
        end = end > c + e + 1 ? end : c + e + 1;
      }
      if (neutral) nbits |= 1u << e;
    }
    if (nbits != 0u && m.words) __hip_atomic_fetch_or(&bits[c >> 5], nbits << (c & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  mask_bounds_finish(m, bits, first, end);
}

}  // namespace ffpa
