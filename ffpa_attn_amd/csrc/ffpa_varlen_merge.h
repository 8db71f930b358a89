// ffpa_varlen_merge.h — stage 2 of a KV-split PACKED-SEQUENCE launch (ffpa_attn_varlen_fwd with FwdArgs::nsplit > 1): combine the normalised fp32 partials the
// ranges of ffpa_fwd_m16_varlen_kernel stored, O = sum_s w_s O_s / sum_s w_s with w_s = exp(LSE_s - max_s LSE_s), LSE = max + ln(sum_s w_s) — the reference's decode
// stage 2 (csrc/cuffpa/native/sm_80/split_kv.cuh:329-455), as ffpa_fwd_merge_kernel (ffpa_fwd_kernel.h) does for the dense call.  What differs from that kernel:
//   * the layouts are the packed call's: partials [split, query head, token, Dk] fp32 + LSE [split, query head, token]; O [token, head, D] by (token, head) strides;
//     LSE [head, lse_stride_h];
//   * a row no range saw a key for (every partial LSE = -inf: an empty key range; a sequence shorter than its query rows under the causal flag) comes out as O = 0 /
//     LSE = -inf — the packed entry point's contract (tests/test_ffpa_cute_sm100.py:1117-1183) — where the dense merge keeps SDPA's NaN;
//   * tokens at and past cu_seqlens_q[batch] belong to no sequence: nothing was stored for them, nothing is written.
// One 64-lane workgroup per (row, 256-column chunk).  Included by ffpa_capi.hip only (the kernels' TUs do not see it).
#pragma once

#include "ffpa_fwd_m16_kernel.h"

namespace ffpa {

template <typename T>
__global__ __launch_bounds__(64) void ffpa_varlen_merge_kernel(const FwdArgs a, const VarlenArgs va, int D, int batch, int64_t o_head_stride) {
  __shared__ float wsh[kMergeMaxSplits];
  const int64_t row = blockIdx.x;  // head * total_q + token
  const int hq = (int)(row / va.ws_head_rows);
  const int tok = (int)(row - (int64_t)hq * va.ws_head_rows);
  if (tok >= va.cu_q[batch]) return;
  const int lane = threadIdx.x;
  float mx = -INFINITY;
  for (int s = lane; s < a.nsplit; s += 64) mx = fmaxf(mx, a.ws_lse[s * va.ws_split_rows + row]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float wsum = 0.f;
  for (int s = lane; s < a.nsplit; s += 64) {
    const float w = (mx == -INFINITY) ? 0.f : __expf(a.ws_lse[s * va.ws_split_rows + row] - mx);
    wsh[s] = w;
    wsum += w;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wsum += __shfl_xor(wsum, o);
  __syncthreads();
  const float inv = wsum > 0.f ? 1.f / wsum : 0.f;  // no visible key in any range: O = 0 (the dense merge: NaN)
  const int d = blockIdx.y * 256 + lane * 4;
  if (d < a.d_valid) {  // D = the kernel's (64-multiple) head dim = the partials' row length; only the caller's columns are stored
    const float* src = a.ws_o + row * D + d;
    const int64_t sstride = va.ws_split_rows * D;
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
    T* op = (T*)a.o + (int64_t)tok * va.o_tok_stride + (int64_t)hq * o_head_stride;
    typename Elem<T>::v4 w4;
#pragma unroll
    for (int e = 0; e < 4; ++e) w4[e] = (T)(acc[e] * inv);
    *(typename Elem<T>::v4*)(op + d) = w4;
  }
  if (a.lse != nullptr && lane == 0 && blockIdx.y == 0) a.lse[(int64_t)hq * va.lse_stride_h + tok] = (mx == -INFINITY) ? -INFINITY : mx + __logf(wsum);
}

}  // namespace ffpa
