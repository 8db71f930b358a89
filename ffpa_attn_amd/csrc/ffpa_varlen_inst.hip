// ffpa_varlen_inst.hip — the packed-sequence kernel (ffpa_fwd_m16_varlen_kernel, ffpa_fwd_m16_kernel.h), one translation unit per head dim
// (compiled with -DFFPA_INST_D=<D>, D a multiple of 64 in [128, 1024]; bf16 + fp16 in the same TU).  A TU of its own so that the dense kernels'
// objects (ffpa_fwd_inst.hip) are exactly what they were before this entry point existed.  What it replaces in the reference: the CuTe-DSL
// launchers behind torch.ops.ffpa_attn._varlen_fwd_cute (src/ffpa_attn/cute/__init__.py:792-829).
#include <atomic>

#include "ffpa_fwd_kernel.h"
#include "ffpa_fwd_m16_kernel.h"
#include "ffpa_launch.h"

#ifndef FFPA_INST_D
#error "compile with -DFFPA_INST_D=<head dim>"
#endif

namespace ffpa {

template <typename T, int D, bool NT>
static int launch_varlen(const FwdArgs& a, const VarlenArgs& va, hipStream_t stream) {
  constexpr int BC = m16_block_keys(D, false);
  constexpr int LDS = 2 * BC * D * 2 + m16_exchange_bytes(D, 0);
  auto kern = ffpa_fwd_m16_varlen_kernel<T, D, NT>;
  static std::atomic<bool> attr_done[64];  // write-once per device (setting the attribute twice is harmless)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (dev < 0 || dev >= 64 || !attr_done[dev].load(std::memory_order_acquire)) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      (void)hipGetLastError();
      return -2;
    }
    if (dev >= 0 && dev < 64) attr_done[dev].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)a.total_wg), dim3(256), LDS, stream, a, va);
  return (int)hipGetLastError();
}

#define FFPA_CAT2(a, b) a##b
#define FFPA_CAT(a, b) FFPA_CAT2(a, b)

// nt: the decode-batch build (K / V pieces with the non-temporal hint: ffpa_capi.hip decides per launch)
int FFPA_CAT(launch_varlen_d, FFPA_INST_D)(int dtype, int nt, const FwdArgs& a, const VarlenArgs& va, hipStream_t stream) {
  if (dtype == 0) return nt ? launch_varlen<__bf16, FFPA_INST_D, true>(a, va, stream) : launch_varlen<__bf16, FFPA_INST_D, false>(a, va, stream);
  if (dtype == 1) return nt ? launch_varlen<_Float16, FFPA_INST_D, true>(a, va, stream) : launch_varlen<_Float16, FFPA_INST_D, false>(a, va, stream);
  return -4;
}

}  // namespace ffpa
