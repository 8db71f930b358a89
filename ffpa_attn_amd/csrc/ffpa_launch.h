// ffpa_launch.h — per-head-dim launch entry points (defined in ffpa_fwd_inst.hip,
// one object per D) and the table the C-ABI dispatches through.
#pragma once
#include <hip/hip_runtime.h>

namespace ffpa {
struct FwdArgs;

#define FFPA_FOR_EACH_HEAD_DIM(X) \
  X(64) X(128) X(192) X(256) X(320) X(384) X(448) X(512) \
  X(576) X(640) X(704) X(768) X(832) X(896) X(960) X(1024)

#define FFPA_DECL(D)                                                                  \
  int launch_fwd_d##D(int dtype, int safe, int variant, const FwdArgs& a, hipStream_t stream); \
  void tile_config_d##D(int variant, int* br, int* bc, int* lds);
FFPA_FOR_EACH_HEAD_DIM(FFPA_DECL)
#undef FFPA_DECL

// the packed-sequence kernel (ffpa_varlen_inst.hip, one object per D): head dims of the 16x16x32 build; smaller ones run on the first of them
struct VarlenArgs;
#define FFPA_FOR_EACH_VARLEN_HEAD_DIM(X) \
  X(128) X(192) X(256) X(320) X(384) X(448) X(512) \
  X(576) X(640) X(704) X(768) X(832) X(896) X(960) X(1024)
#define FFPA_DECL(D) int launch_varlen_d##D(int dtype, int nt, const FwdArgs& a, const VarlenArgs& va, hipStream_t stream);
FFPA_FOR_EACH_VARLEN_HEAD_DIM(FFPA_DECL)
#undef FFPA_DECL

}  // namespace ffpa
