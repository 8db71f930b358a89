// ffpa_fwd_inst.hip — one translation unit per head dim (compiled with
// -DFFPA_INST_D=<D>); keeps hipcc invocations small and parallel.  The reference
// generates one TU per (dtype, acc, headdim, stage) from env.py:455-521; here the
// only axis is the head dim (bf16 + fp16 in the same TU).
#include <atomic>

#include "ffpa_fwd_kernel.h"
#include "ffpa_fwd_m16_kernel.h"
#include "ffpa_launch.h"

#ifndef FFPA_INST_D
#error "compile with -DFFPA_INST_D=<head dim>"
#endif

namespace ffpa {

template <typename T, int D, int ND, bool SAFE, bool DROP = false, bool BTILE = false, int MK = 1>
static int launch_one(const FwdArgs& a, hipStream_t stream) {
  constexpr int BC = (ND == 1) ? ((D <= FFPA_BC128_MAX_D && !BTILE) ? 128 : 64) : 32;
  constexpr int LDS_BASE = 2 * BC * D * 2 + (ND > 1 ? 4 * 4096 : 0);
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
#if FFPA_PERSISTENT
  // one workgroup per CU (this kernel's LDS / register footprint admits exactly one), each walking total / grid ids
  if (ND <= 2 && a.nsplit == 1 && !(a.flags & 0x4u)) {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned g = (unsigned)(cus > 8 ? cus - cus % 8 : 8);
    if (grid > g) grid = g;
  }
#endif
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, stream, a);
  return (int)hipGetLastError();
}

// The unmasked / boolean-mask prefill kernel on the 16x16x32 MFMA shape (ffpa_fwd_m16_kernel.h): same tiles, same plan.
template <typename T, int D, int MK, bool DROP = false>
static int launch_m16(const FwdArgs& a, hipStream_t stream) {
  constexpr int LDS_BASE = D > 512 ? 2 * 32 * D * 2 + 4 * 4096 : 2 * ((D <= FFPA_BC128_MAX_D) ? 128 : 64) * D * 2;
  const int LDS = LDS_BASE + (a.bias_lds > 0 ? a.bias_lds : 0);  // + the key-bias row cache, sized by the C-ABI layer
  auto kern = ffpa_fwd_m16_kernel<T, D, MK, DROP>;
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

#define FFPA_CAT2(a, b) a##b
#define FFPA_CAT(a, b) FFPA_CAT2(a, b)

int FFPA_CAT(launch_fwd_d, FFPA_INST_D)(int dtype, int safe, int variant, const FwdArgs& a, hipStream_t stream) {
  constexpr int D = FFPA_INST_D;
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
  if (a.dropout_p > 0.f) {
    if constexpr (D >= FFPA_M16_MIN_D) {
      if (!(a.flags & 0x10u)) {
        if (dtype == 0) return launch_m16<__bf16, D, 1, true>(a, stream);
        if (dtype == 1) return launch_m16<_Float16, D, 1, true>(a, stream);
        return -4;
      }
    }
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
  if (a.bias_dtype == 0 && a.kv_bounds == nullptr) {  // no attn_bias, no mask ranges: the build without any bias path
    if constexpr (D >= FFPA_M16_MIN_D) {
      if (!(a.flags & 0x10u) && a.scale_log2 > 0.f) {  // (FFPA_FLAG_NO_M16 keeps the 32x32x16 build: A/B runs, tests; so do scales <= 0: this build folds the scale into the exponent's FMA)
        if (dtype == 0) return launch_m16<__bf16, D, 0>(a, stream);
        if (dtype == 1) return launch_m16<_Float16, D, 0>(a, stream);
        return -4;
      }
    }
    if (dtype == 0) return launch_one<__bf16, D, ND, false, false, false, 0>(a, stream);
    if (dtype == 1) return launch_one<_Float16, D, ND, false, false, false, 0>(a, stream);
    return -4;
  }
  if (a.bias_dtype == 4) {  // boolean mask (+ ranges): the build that carries only that path
    if constexpr (D >= FFPA_M16_MIN_D) {
      if (!(a.flags & 0x10u) && a.scale_log2 > 0.f) {
        if (dtype == 0) return launch_m16<__bf16, D, 2>(a, stream);
        if (dtype == 1) return launch_m16<_Float16, D, 2>(a, stream);
        return -4;
      }
    }
    if (dtype == 0) return launch_one<__bf16, D, ND, false, false, false, 2>(a, stream);
    if (dtype == 1) return launch_one<_Float16, D, ND, false, false, false, 2>(a, stream);
    return -4;
  }
  // (additive biases stay on the 32x32x16 build: its lanes hold 16 consecutive keys per block — two 16-byte bias loads — where the
  // 16x16x32 layout holds 4 keys of two rows; measured 5 ... 25 % slower there, tools/gpu_bias_m16_ab.py)
  if (dtype == 0) return launch_one<__bf16, D, ND, false>(a, stream);
  if (dtype == 1) return launch_one<_Float16, D, ND, false>(a, stream);
  return -4;
}

void FFPA_CAT(tile_config_d, FFPA_INST_D)(int variant, int* br, int* bc, int* lds) {
  constexpr int D = FFPA_INST_D;
  // variant 0: prefill tiles, 1: short-query tiles, 2: prefill tiles of the bias-tile build (64 keys at every head dim)
  const int ND = variant == 1 ? ((D % 128 == 0) ? 4 : 2) : ((D <= 512) ? 1 : 2);
  const int BC = (ND == 1) ? ((D <= FFPA_BC128_MAX_D && variant != 2) ? 128 : 64) : 32;
  *br = 32 * (4 / ND);
  *bc = BC;
  *lds = 2 * BC * D * 2 + (ND > 1 ? 4 * 4096 : 0);
}

}  // namespace ffpa
