// ffpa_fwd_inst.hip — one translation unit per head dim (compiled with
// -DFFPA_INST_D=<D>); keeps hipcc invocations small and parallel.  The reference
// generates one TU per (dtype, acc, headdim, stage) from env.py:455-521; here the
// only axis is the head dim (bf16 + fp16 in the same TU).
#include <atomic>

#include "ffpa_fwd_kernel.h"
#include "ffpa_fwd_m16_kernel.h"
#include "ffpa_fwd_m16w_kernel.h"
#include "ffpa_launch.h"

#ifndef FFPA_INST_D
#error "compile with -DFFPA_INST_D=<head dim>"
#endif

namespace ffpa {

template <typename T, int D, int ND, bool SAFE, bool DROP = false, bool BTILE = false, int MK = 1>
static int launch_one(const FwdArgs& a, hipStream_t stream) {
  constexpr int BC = (ND == 1) ? ((D <= FFPA_BC128_MAX_D && !BTILE) ? 128 : 64) : splitd_block_keys(D, ND);
  constexpr int LDS_BASE = 2 * BC * D * 2 + splitd_exchange_bytes(D, ND);
  const int LDS = LDS_BASE + (a.bias_lds > 0 ? a.bias_lds : -a.bias_lds);  // + the key-bias row cache or the bias-tile staging area, sized by the C-ABI layer (<= 160 KiB in total)
  constexpr int kMaxLds = 160 * 1024;
  auto kern = ffpa_fwd_split_d_kernel<T, D, ND, SAFE, DROP, BTILE, MK>;
  static std::atomic<bool> attr_done[64];  // write-once per device (setting the attribute twice is harmless)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (dev < 0 || dev >= 64 || !attr_done[dev].load(std::memory_order_acquire)) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds) != hipSuccess) {
      (void)hipGetLastError();
      return -2;
    }
    if (dev >= 0 && dev < 64) attr_done[dev].store(true, std::memory_order_release);
  }
  unsigned grid = (unsigned)a.total_wg;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, stream, a);
  return (int)hipGetLastError();
}

// The prefill kernel on the 16x16x32 MFMA shape (ffpa_fwd_m16_kernel.h): every prefill launch at head dims >= FFPA_M16_MIN_D.
template <typename T, int D, int MK, bool DROP = false, bool PAIR = false>
static int launch_m16(const FwdArgs& a, hipStream_t stream) {
  if constexpr (MK == 0 && !PAIR) {
    if (a.pair_tiles) return launch_m16<T, D, MK, DROP, true>(a, stream);  // (the kernel whose workgroups walk two row tiles: causal launches, ffpa_capi.hip::pick_pair_tiles)
  }
  constexpr int BC = m16_block_keys(D, MK == 1 || MK == 3);
  constexpr int LDS_BASE = 2 * BC * D * 2 + m16_exchange_bytes(D, MK);
  const int LDS = LDS_BASE + (a.bias_lds > 0 ? a.bias_lds : -a.bias_lds);  // + the key-bias row cache or the bias-tile staging areas, sized by the C-ABI layer (<= 160 KiB in total)
  void (*kern)(const FwdArgs);
  if constexpr (PAIR) kern = ffpa_fwd_m16_pair_kernel<T, D, DROP>;
  else kern = ffpa_fwd_m16_kernel<T, D, MK, DROP>;
  static std::atomic<bool> attr_done[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (dev < 0 || dev >= 64 || !attr_done[dev].load(std::memory_order_acquire)) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      (void)hipGetLastError();
      return -2;
    }
    if (dev >= 0 && dev < 64) attr_done[dev].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)a.total_wg), dim3(256), LDS, stream, a);
  return (int)hipGetLastError();
}

// The wide-row prefill tile of the head dims whose O^T leaves accumulator registers idle at 32 rows per wave (ffpa_fwd_m16w_kernel.h).
template <typename T, int D, int MK>
static int launch_m16w(const FwdArgs& a, hipStream_t stream) {
  if constexpr (m16w_available(D)) {
    constexpr int RH = m16w_row_halves(D);
    auto kern = ffpa_fwd_m16w_kernel<T, D, RH, m16w_block_keys(D), MK>;
    static std::atomic<bool> attr_done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (dev < 0 || dev >= 64 || !attr_done[dev].load(std::memory_order_acquire)) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        (void)hipGetLastError();
        return -2;
      }
      if (dev >= 0 && dev < 64) attr_done[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)a.total_wg), dim3(256), m16w_lds_bytes(D), stream, a);
    return (int)hipGetLastError();
  } else {
    return -3;
  }
}

#define FFPA_CAT2(a, b) a##b
#define FFPA_CAT(a, b) FFPA_CAT2(a, b)

// Which kernel a launch runs (also reported by ffpa_attn_fwd_plan: FFPA_KERNEL_* in include/ffpa_attn.h):
//   short-query tiles (variant 1)           -> ffpa_fwd_split_d_kernel<ND = 4 / 2>  (32x32x16 MFMA, split-KV)
//   prefill, head dim >= FFPA_M16_MIN_D     -> ffpa_fwd_m16_kernel<MK, DROP>        (16x16x32 MFMA): the one prefill family of the large head dims
//                                              (MK: 0 no bias, 2 boolean mask / ranges, 3 key bias from the LDS row cache, 1 any other additive bias or mask)
//   prefill, smaller head dims              -> ffpa_fwd_split_d_kernel<ND = 1>      (32x32x16 MFMA)
// (a template, so that `if constexpr` really discards the other family's instantiations)
template <int D>
static int launch_fwd_impl(int dtype, int safe, int variant, const FwdArgs& a, hipStream_t stream) {
  constexpr int ND = (D <= 512) ? 1 : 2;
  if (variant == 1) {
    // short-query launches: D split over all 4 waves (one 32-row block per workgroup) when the D/4
    // slice is a whole number of 32-column O blocks, else over 2 waves (two row blocks)
    constexpr int NDS = (D % 128 == 0) ? 4 : 2;
    if (safe) return -3;
    if (a.dropout_p > 0.f) {
      if (dtype == 0) return launch_one<__bf16, D, NDS, false, true>(a, stream);
      if (dtype == 1) return launch_one<_Float16, D, NDS, false, true>(a, stream);
      return -4;
    }
    if (dtype == 0) return launch_one<__bf16, D, NDS, false>(a, stream);
    if (dtype == 1) return launch_one<_Float16, D, NDS, false>(a, stream);
    return -4;
  }
#ifdef FFPA_INST_SAFE
  if (safe) {
    if (a.dropout_p > 0.f) return -3;
    if (dtype == 0) return launch_one<__bf16, D, ND, true>(a, stream);
    return -3;
  }
#else
  if (safe) return -3;
#endif
  const bool no_bias = a.bias_dtype == 0 && a.kv_bounds == nullptr;  // no attn_bias, no mask ranges: the builds without any bias path
  if (variant == 3) {  // the wide-row tile (the plan only picks it for the builds that exist: no additive bias, no dropout)
    if (a.dropout_p > 0.f || !(a.bias_dtype == 0 || a.bias_dtype == 4)) return -3;
    if (no_bias) {
      if (dtype == 0) return launch_m16w<__bf16, D, 0>(a, stream);
      if (dtype == 1) return launch_m16w<_Float16, D, 0>(a, stream);
      return -4;
    }
    if (dtype == 0) return launch_m16w<__bf16, D, 2>(a, stream);
    if (dtype == 1) return launch_m16w<_Float16, D, 2>(a, stream);
    return -4;
  }
  if constexpr (D >= FFPA_M16_MIN_D) {
    if (a.dropout_p > 0.f) {
      if (no_bias) {
        if (dtype == 0) return launch_m16<__bf16, D, 0, true>(a, stream);
        if (dtype == 1) return launch_m16<_Float16, D, 0, true>(a, stream);
        return -4;
      }
      if (dtype == 0) return launch_m16<__bf16, D, 1, true>(a, stream);
      if (dtype == 1) return launch_m16<_Float16, D, 1, true>(a, stream);
      return -4;
    }
    if (no_bias) {
      if (dtype == 0) return launch_m16<__bf16, D, 0>(a, stream);
      if (dtype == 1) return launch_m16<_Float16, D, 0>(a, stream);
      return -4;
    }
    if (a.bias_dtype == 4 || a.bias_dtype == 0) {  // boolean mask and / or mask ranges: the build that carries only that path
      if (dtype == 0) return launch_m16<__bf16, D, 2>(a, stream);
      if (dtype == 1) return launch_m16<_Float16, D, 2>(a, stream);
      return -4;
    }
    if (a.bias_lds > 0 && a.kv_bounds == nullptr) {  // a key bias that fits the LDS row cache, nothing else: the lean key-bias build
      if (dtype == 0) return launch_m16<__bf16, D, 3>(a, stream);
      if (dtype == 1) return launch_m16<_Float16, D, 3>(a, stream);
      return -4;
    }
    if (dtype == 0) return launch_m16<__bf16, D, 1>(a, stream);
    if (dtype == 1) return launch_m16<_Float16, D, 1>(a, stream);
    return -4;
  } else {
    if (a.dropout_p > 0.f) {
      if (dtype == 0) return launch_one<__bf16, D, ND, false, true>(a, stream);
      if (dtype == 1) return launch_one<_Float16, D, ND, false, true>(a, stream);
      return -4;
    }
    // 16-bit bias with a row axis, staged through LDS one step ahead (the build with 64-key tiles at every head dim, so that the
    // LDS holds the bias tiles next to K and V; the plan of the C-ABI layer uses tile_config variant 2 for these launches)
    if (a.bias_tile) {
      if (dtype == 0) return launch_one<__bf16, D, ND, false, false, true>(a, stream);
      if (dtype == 1) return launch_one<_Float16, D, ND, false, false, true>(a, stream);
      return -4;
    }
    if (no_bias) {
      if (dtype == 0) return launch_one<__bf16, D, ND, false, false, false, 0>(a, stream);
      if (dtype == 1) return launch_one<_Float16, D, ND, false, false, false, 0>(a, stream);
      return -4;
    }
    if (a.bias_dtype == 4 || a.bias_dtype == 0) {  // boolean mask (+ ranges): the build that carries only that path
      if (dtype == 0) return launch_one<__bf16, D, ND, false, false, false, 2>(a, stream);
      if (dtype == 1) return launch_one<_Float16, D, ND, false, false, false, 2>(a, stream);
      return -4;
    }
    if (dtype == 0) return launch_one<__bf16, D, ND, false>(a, stream);
    if (dtype == 1) return launch_one<_Float16, D, ND, false>(a, stream);
    return -4;
  }
}

int FFPA_CAT(launch_fwd_d, FFPA_INST_D)(int dtype, int safe, int variant, const FwdArgs& a, hipStream_t stream) {
  return launch_fwd_impl<FFPA_INST_D>(dtype, safe, variant, a, stream);
}

void FFPA_CAT(tile_config_d, FFPA_INST_D)(int variant, int* br, int* bc, int* lds) {
  constexpr int D = FFPA_INST_D;
  // variant 0: prefill tiles, 1: short-query tiles, 2: prefill tiles of the additive-bias builds (64 keys at every head dim <= 512:
  // the 16x16x32 build with any additive bias, the 32x32x16 build with LDS-staged bias tiles), 3: the wide-row prefill tile
  // (ffpa_fwd_m16w_kernel.h; *br = 0 where the head dim has none), 4: the key-bias build of the split-D tiles where it runs the softmax pipeline (the tiles of
  // variant 0; its LDS without the ring cache)
  if (variant == 3) {
    *br = m16w_available(D) ? m16w_block_rows(m16w_row_halves(D)) : 0;
    *bc = m16w_block_keys(D);
    *lds = m16w_available(D) ? m16w_lds_bytes(D) : 0;
    return;
  }
  const int ND = variant == 1 ? ((D % 128 == 0) ? 4 : 2) : ((D <= 512) ? 1 : 2);
  int BC = (ND == 1) ? ((D <= FFPA_BC128_MAX_D && variant != 2) ? 128 : 64) : splitd_block_keys(D, ND);
  if (variant != 1 && D >= FFPA_M16_MIN_D) BC = m16_block_keys(D, variant == 2);  // (the 16x16x32 kernel's own rule)
  *br = 32 * (4 / ND);
  *bc = BC;
  *lds = 2 * BC * D * 2 + (variant == 1 ? splitd_exchange_bytes(D, ND) : (D >= FFPA_M16_MIN_D ? m16_exchange_bytes(D, variant == 2 ? 1 : (variant == 4 ? 3 : 0)) : (ND > 1 ? 4 * 4096 : 0)));
}

}  // namespace ffpa
