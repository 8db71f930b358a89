/*
 * ffpa_attn.h — C-ABI of the MI355X (gfx950) fused attention forward.
 *
 * This is the drop-in boundary for the ONE hot path this repo accelerates:
 * the large-headdim Split-D attention forward that the reference reaches via
 *
 *   torch.ops.ffpa_attn._fwd_cuda            (src/ffpa_attn/cuda/__init__.py:57-139)
 *     -> ffpa_attn._C.ffpa_attn_forward      (csrc/cuffpa/ffpa_api.cc:86-239, pybind :265-306)
 *       -> launch_ffpa_attn_fwd_template     (csrc/cuffpa/launch.cuh:61-606)
 *         -> split_d_fwd_sm80 & friends      (csrc/cuffpa/native/sm_80/split_d.cuh:96-777)
 *
 * Every entry point is extern "C", takes plain pointers / sizes / strides and
 * a hipStream_t passed as void*.  No torch types, no allocation, no device
 * synchronisation, no global mutable state beyond write-once per-device caches (CU count, arch check,
 * kernel attributes; std::atomic): the caller owns every buffer and
 * picks the stream.  Each function cites the reference interface it replaces.
 */
#ifndef FFPA_ATTN_H_
#define FFPA_ATTN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFPA_ATTN_ABI_VERSION 6 /* 5: + the packed-sequence entry points (ffpa_attn_varlen_fwd ...); 6: + KV splits inside the packed call (its workspace fields, ffpa_attn_varlen_fwd_workspace_bytes, plan out[4]) */

/* status codes (0 == success).  The Python host maps them onto the exception
 * classes the reference raises (TORCH_CHECK -> RuntimeError,
 * std::invalid_argument -> ValueError; ffpa_api.cc:193-197, env.py:750-752). */
enum ffpa_status {
  FFPA_OK = 0,
  FFPA_ERR_NULL_POINTER = 1,      /* q/k/v/o missing                                   */
  FFPA_ERR_BAD_DTYPE = 2,         /* dtype not bf16/fp16, or bias dtype unknown         */
  FFPA_ERR_BAD_HEADDIM = 3,       /* "headdim not support!" (env.py:750-752)            */
  FFPA_ERR_BAD_SHAPE = 4,         /* non-positive dims, Hq % Hkv != 0                   */
  FFPA_ERR_BAD_STRIDE = 5,        /* stride not a multiple of 8 elements / too large    */
  FFPA_ERR_MISALIGNED = 6,        /* base pointer not 16-byte aligned                   */
  FFPA_ERR_UNSUPPORTED = 7,       /* feature not built (e.g. dropout)                   */
  FFPA_ERR_LAUNCH = 8,            /* hipGetLastError() after the launch                 */
  FFPA_ERR_NO_DEVICE = 9,         /* no current device, or it is not a gfx950           */
  FFPA_ERR_BAD_ABI = 10           /* struct_size / abi_version mismatch                 */
};

enum ffpa_dtype { FFPA_DTYPE_BF16 = 0, FFPA_DTYPE_FP16 = 1 };

/* attn_bias element type.  Same codes as the reference's native launcher
 * (csrc/cuffpa/native/launch.cuh:279-281): 0 none, 1 fp16, 2 bf16, 3 fp32 — plus 4: a boolean mask, one byte per
 * score, non-zero = the key is visible, zero = score -inf.  The reference's host materialises that as a 0 / -inf
 * tensor in q's dtype before every launch (src/ffpa_attn/functional.py:891-898); here the kernel reads the caller's
 * bytes (torch.bool storage) directly: same scores, no mask-sized temporary, 1 byte per element of mask traffic. */
enum ffpa_bias_dtype {
  FFPA_BIAS_NONE = 0,
  FFPA_BIAS_FP16 = 1,
  FFPA_BIAS_BF16 = 2,
  FFPA_BIAS_FP32 = 3,
  FFPA_BIAS_BOOL8 = 4
};

/* ffpa_fwd_params.flags */
#define FFPA_FLAG_DEBUG_SAFE_PATH 0x1u /* test-only: register-staged K/V + scalar V gather  */
#define FFPA_FLAG_NO_XCD_REMAP    0x2u /* bench-only: dispatch-order block mapping          */
/* (0x4u: retired — it belonged to the persistent-workgroup experiment, tools/experiments/r05_pruned_switches.diff) */
#define FFPA_FLAG_NO_BIAS_LDS     0x8u /* bench-only: read a key bias from global memory in every tile   */
#define FFPA_FLAG_L2_PREFETCH     0x10u /* bench-only: touch the K/V tile two steps ahead in every prefill launch (default: the launch side decides) */
#define FFPA_FLAG_NO_L2_PREFETCH  0x20u /* bench-only: never                                           */
#define FFPA_FLAG_FORCE_SPLITS    0x40u /* bench-only: honour num_splits > 1 for a prefill launch that fills the chip too (default: only under-filled launches split) */
#define FFPA_FLAG_KV_STREAM       0x80u /* bench-only: short-query launches fetch K / V with the non-temporal hint (default: the launch side decides — one reader per byte and K + V larger than the Infinity Cache) */
#define FFPA_FLAG_NO_KV_STREAM    0x800u /* bench-only: never                                          */
#define FFPA_FLAG_WIDE_TILE       0x1000u /* bench / test: prefill launches take the wide-row tile (ffpa_fwd_m16w_kernel) wherever the head dim and the mask kind have one (default: the launch side decides) */
#define FFPA_FLAG_NO_WIDE_TILE    0x2000u /* bench / test: never                                       */
#define FFPA_FLAG_PAIR_TILES      0x8000u /* bench / test: causal prefill launches pair row tiles i and n - 1 - i in one workgroup wherever the build can (default: the launch side decides) */
#define FFPA_FLAG_NO_PAIR_TILES   0x20000u /* bench / test: never                                      */
#define FFPA_FLAG_DETERMINISTIC   0x4000u /* batch-invariant bits: the plan of a (batch, head) slice does not depend on how many slices share the launch — prefill
                                             launches never split the KV axis and never take the wide-row tile (128-row tiles, one pass per row), short-query launches
                                             split by the KV length alone (a fixed number of KV tiles per range).  Costs what the launch-size rules would have gained
                                             (under-filled / ragged-round prefill launches: up to ~ 20 %).  Python: FFPA_HIP_DETERMINISTIC=1 sets it on every call. */
#define FFPA_FLAG_NO_HEAD_CHUNKS   0x80000u /* bench / test, ffpa_attn_fwd: causal GQA prefill launches keep the (batch, head, row tile) workgroup order (default: the launch side takes the head-chunk order — the same row tile of a KV group's heads at the same time — where its rule applies; same bits either way) */
#define FFPA_FLAG_TILE_RANGES      0x100000u /* bench / test, ffpa_attn_fwd: with FFPA_FLAG_FORCE_SPLITS and num_splits = n, a causal prefill launch splits every row tile's OWN visible KV tiles into n ranges (the packed-sequence kernel's dense mode) instead of n uniform ranges (default: the launch side takes two such ranges for causal launches of one round of workgroups) */
#define FFPA_FLAG_NO_TILE_RANGES   0x200000u /* bench / test, ffpa_attn_fwd: never */
#define FFPA_FLAG_NO_COMPACT_GRID   0x400000u /* bench / test, ffpa_attn_varlen_fwd: a ragged prefill batch keeps the grid of batch x ceil(max_seqlen_q / block rows) row tiles per head (default: callers that say total_q get ceil(total_q / block rows) + batch slots per head when three quarters of the full grid would be idle; same order, same bits) */
#define FFPA_FLAG_NO_PACK_GQA      0x40000u /* bench / test, ffpa_attn_varlen_fwd: short query sequences under GQA (group x max_seqlen_q rows fit one tile: decode, speculative decoding) keep one workgroup per QUERY head (default: the heads of a KV group x the tokens are packed into the rows of one tile) */
#define FFPA_FLAG_XCD_GROUP(log2p1) ((unsigned)(log2p1) << 8) /* bench-only: bits 8..10 = 1 + log2 of the XCDs that share a head's row tiles (1 -> 1, 2 -> 2, 3 -> 4, 4 -> 8); 0 = the launch side decides */

/*
 * One forward call.  Layout contract (replaces the dense-[B,H,N,D] assumption of
 * split_d.cuh:137-142): element strides for the batch / head / sequence dims,
 * headdim stride 1, every row 16-byte aligned (D % 8 == 0, strides % 8 == 0).
 *
 *   q  [B, Hq,  Nq,  D]     k, v [B, Hkv, Nkv, D]     o [B, Hq, Nq, D]
 *   lse  [B, Hq, Nq] fp32 contiguous, natural log, may be NULL (cuda/__init__.py:102-112)
 *   bias [B|1, Hq|1, Nq|1, Nkv|1] additive (fp16 / bf16 / fp32) or boolean (FFPA_BIAS_BOOL8), stride 0 on
 *        broadcast dims (native/launch.cuh:277-290); NULL <=> bias_dtype == FFPA_BIAS_NONE.
 *
 * Score = scale * q.k + bias ; masked iff (causal && key > row + causal_offset)
 * or key >= Nkv or (boolean mask && mask byte == 0).  causal_offset = Nkv - Nq reproduces the reference's
 * tail-aligned causal mask (split_d.cuh:222-228); 0 reproduces PyTorch SDPA's
 * top-left alignment (SURVEY.md §8 "config-4 semantic trap").
 */
typedef struct ffpa_fwd_params {
  uint32_t struct_size; /* sizeof(ffpa_fwd_params), checked */
  uint32_t abi_version; /* FFPA_ATTN_ABI_VERSION            */

  const void* q;
  const void* k;
  const void* v;
  void* o;
  float* lse;       /* optional */
  const void* bias; /* optional */

  int32_t batch;       /* B   */
  int32_t heads_q;     /* Hq  */
  int32_t heads_kv;    /* Hkv */
  int32_t seqlen_q;    /* Nq  */
  int32_t seqlen_kv;   /* Nkv */
  int32_t head_dim;    /* D: any multiple of 8 in [8, 1024] */

  int64_t q_stride[3];    /* elements: batch, head, row */
  int64_t k_stride[3];
  int64_t v_stride[3];
  int64_t o_stride[3];
  int64_t bias_stride[4]; /* elements: batch, head, row, key (0 = broadcast) */

  int32_t dtype;         /* enum ffpa_dtype      */
  int32_t bias_dtype;    /* enum ffpa_bias_dtype */
  int32_t causal;        /* 0 / 1                */
  int32_t causal_offset; /* visible iff key <= row + causal_offset */

  float softmax_scale;     /* applied to q.k before softmax                       */
  float rescale_threshold; /* lazy-rescale threshold in log2 units; <0 => default
                              8.0 (FFPA_RESCALE_THRESHOLD, csrc/cuffpa/common.cuh:14);
                              0 => exact recurrence                                */
  float dropout_p;         /* [0, 1): P <- keep ? P/(1-p) : 0 after the row sum (prefill.cuh:508-546) */
  uint32_t flags;

  /* Philox4x32-10 key and base counter: element (b,hq,q,k) uses word e&3 of block e>>2 with
   * e = philox_offset + ((b*Hq + hq)*Nq + q)*Nkv + k; keep iff (word+1)*2^-32 > dropout_p — the
   * SDPA-efficient-attention-compatible convention of prefill.cuh:398-452.  The caller reserves the
   * ceil4(B*Hq*Nq*Nkv) offsets from its generator (functional.py:518-540). */
  uint64_t philox_seed;
  uint64_t philox_offset;

  /* Short-query (decode) launches, seqlen_q <= 32 — and prefill launches that would fill less than half of
   * the CUs: the KV axis is split over workgroups and merged by
   * LSE, the role of the reference's split_kv_decode_s1/s2 kernels (native/sm_80/split_kv.cuh:22-455,
   * heuristic native/launch.cuh:17-67).  The caller owns the scratch (the reference allocates it inside
   * the launcher, native/launch.cuh:314-318): size from ffpa_attn_fwd_workspace_bytes(). */
  void* workspace;          /* device scratch, 16-byte aligned; NULL => no split        */
  uint64_t workspace_bytes;
  int32_t num_splits;       /* 0 = library heuristic, 1 = never split, n = at most n     */
  /* Rows of several query heads of one KV group packed into the row axis by the caller (q viewed as
   * [B, Hkv, group*Nq, D]): the causal limit of packed row r is (r % causal_row_mod) + causal_offset.
   * 0 = rows are plain query rows. */
  int32_t causal_row_mod;

  /* Optional key ranges derived from the mask by the caller (NULL = none), four int32 per block of 32 query rows:
   *   {first, end}            EVERY key outside [first, end) is masked (bias == -inf / mask == False) for EVERY row of the
   *                           block; an empty block is {Nkv, 0};
   *   {free_first, free_end}  for EVERY key inside [free_first, free_end) the mask is neutral (bias == 0 / mask == True) for
   *                           EVERY row of the block; {0, 0} = no such claim.
   * Shape [Bb, Hb, ceil(Nq / 32), 4] with element strides kv_bounds_stride = {batch, head} (0 = broadcast);
   * ffpa_attn_mask_kv_bounds() computes it.  The kernel clips its KV-tile loop to the union of its blocks' [first, end) and
   * does not read the mask for tiles inside a block's free range: results are unchanged (the skipped tiles contribute
   * exp(-inf) = 0, the skipped mask reads would have added 0); an explicit causal / sliding-window / padding mask stops
   * costing the tiles it hides AND the tiles it leaves fully visible — only the tiles its edge crosses pay for it.  The
   * reference has no counterpart (it walks every tile and adds the bias everywhere: native/sm_80/split_d.cuh:222-228 clips
   * for is_causal only). */
  const int32_t* kv_bounds;
  int64_t kv_bounds_stride[2];

  /* Optional, short-query (seqlen_q <= 32) KV-split launches only (NULL = none; ignored by other launches):
   * ffpa_attn_fwd_split_tickets(params) int32 counters, ZERO on entry.  With them the
   * split partials are merged inside the same launch — the last split of a row tile to arrive (agent-scope release of its partial, one
   * relaxed atomic ticket, agent-scope acquire by the merger) combines all of them by LSE, writes O / LSE and puts its counter back to
   * zero: one launch per call instead of two, the role of the reference's split_kv_decode stage 2 (native/sm_80/split_kv.cuh:329-455)
   * without its second kernel.  The counters must not be shared by launches that may run concurrently (other streams); after a launch
   * completes they are zero again, so one buffer zeroed once serves every later call on that stream, HIP-graph replays included.
   * Without them a second kernel (ffpa_fwd_merge_kernel) follows the first on the same stream: the same numbers.  Which of the two is
   * faster is a measurement (profiles/r03_split_merge.txt), the Python host's default follows it. */
  int32_t* split_tickets;
} ffpa_fwd_params;

/*
 * Launch the fused forward on `stream` (a hipStream_t; NULL = default stream) of
 * the CURRENT device.  Asynchronous; returns an ffpa_status.
 * Replaces: ffpa_attn._C.ffpa_attn_forward (csrc/cuffpa/ffpa_api.cc:86-239).
 */
int ffpa_attn_fwd(const ffpa_fwd_params* params, void* stream);

/*
 * Scratch bytes the call would use with the split count it would choose (0 when it does not split).
 * Replaces the in-launcher allocations of native/launch.cuh:314-318,503-509.
 */
size_t ffpa_attn_fwd_workspace_bytes(const ffpa_fwd_params* params);

/*
 * Number of int32 counters ffpa_fwd_params.split_tickets must hold for this call (one per (batch, query head, row
 * tile): batch * heads_q * ceil(seqlen_q / block rows) — a short-query launch has one row tile per head today, the count does not
 * rely on it; 0 when the call is not a KV-split short-query launch).  The kernel leaves every counter at zero when the launch
 * completes; after a launch that failed or faulted the caller must zero the buffer again before reusing it.
 */
size_t ffpa_attn_fwd_split_tickets(const ffpa_fwd_params* params);

/*
 * The launch plan for `params` (for benches / roofline maths / tests): out[0] = kernel variant
 * (0 = prefill tiles, 1 = short-query tiles), out[1] = query rows per workgroup, out[2] = keys per
 * tile, out[3] = number of KV splits given params->workspace_bytes.  Returns an ffpa_status.
 */
int ffpa_attn_fwd_plan(const ffpa_fwd_params* params, int out[4]);

/*
 * The kernel the launch described by `params` runs, as text ("ffpa_fwd_m16_kernel<bf16, 512, MK=0, DROP=0>", with
 * " + ffpa_fwd_merge_kernel" appended for KV-split launches): what benches put next to their numbers and what a profile's
 * kernel column must show.  Written to buf (n bytes, NUL-terminated).  Returns an ffpa_status.
 */
int ffpa_attn_fwd_kernel(const ffpa_fwd_params* params, char* buf, size_t n);

/*
 * Visible-key bounds of an additive (-inf = hidden) or boolean (0 = hidden) mask, in the layout
 * ffpa_fwd_params.kv_bounds expects: one fused pass over
 * `bias` ([bb, hb, nq|1, nkv|1] with element strides bias_stride, 0 = broadcast; enum ffpa_bias_dtype) writes
 * out[bb][hb][ceil(nq / 32)][4] = {first, end, free_first, free_end} (int32, contiguous) on `stream`.  Returns an
 * ffpa_status.
 */
int ffpa_attn_mask_kv_bounds(const void* bias, int bias_dtype, const int64_t bias_stride[4], int bb, int hb,
                             int nq, int nkv, int32_t* out, void* stream);

/*
 * PACKED SEQUENCES ("varlen", FlashAttention's THD layout) — the reference's ffpa_attn_varlen_func
 * (src/ffpa_attn/ffpa_attn_interface.py:192-279), which it serves through its CuTe-DSL backend only
 * (torch.ops.ffpa_attn._varlen_fwd_cute, src/ffpa_attn/cute/__init__.py:466-575,792-829: NVIDIA SM8x / SM90 / SM100).
 *
 *   q  [total_q, Hq,  D]     k, v [total_k, Hkv, D]     o [total_q, Hq, D]     (element strides: row, head; head-dim stride 1)
 *   lse [Hq, total_q] fp32, natural log, element stride lse_stride_head between heads; may be NULL
 *   cu_seqlens_q / cu_seqlens_kv: int32 [batch + 1] ON THE DEVICE, non-decreasing, [0] == 0: sequence i owns rows
 *        cu_seqlens_q[i] .. cu_seqlens_q[i + 1] of q / o and cu_seqlens_kv[i] .. cu_seqlens_kv[i + 1] of k / v.
 *
 * ONE launch for the whole batch; the boundaries are read by the kernel (nothing is copied to the host, the call never
 * synchronises and captures into a HIP graph): the grid holds ceil(max_seqlen_q / block rows) row tiles per (sequence,
 * head), workgroups whose tile lies past their sequence's last row leave at once.  max_seqlen_q must be >= every
 * sequence's query length (rows past it would not be computed); max_seqlen_kv is only a hint for the launch side.
 * Short query sequences under GQA — decode (max_seqlen_q == 1), speculative decoding, small prefill chunks: (Hq / Hkv) x max_seqlen_q rows
 * fit one tile — run with the query heads of a KV group x the sequence's tokens packed into the rows of one tile per (sequence, KV head):
 * the group's K / V are read once (the reference's pack_gqa, cute/__init__.py:792-829).
 * Per sequence the arithmetic is the dense call's (same tile, same recurrence: bit-identical to ffpa_attn_fwd on that
 * sequence alone under FFPA_FLAG_DETERMINISTIC).  causal = the reference's tail-aligned mask PER SEQUENCE: row r of
 * sequence i sees key j iff j <= r + (Nkv_i - Nq_i).  Rows without a visible key (an empty key range; the first
 * Nq_i - Nkv_i rows of a causal sequence with more queries than keys) get O = 0 and LSE = -inf — the contract the
 * reference's tests pin for this entry point (tests/test_ffpa_cute_sm100.py:1117-1183) — where ffpa_attn_fwd keeps
 * SDPA's NaN.  No attn_bias, no dropout (the reference rejects both here: cute/__init__.py:76-139).
 */
typedef struct ffpa_varlen_fwd_params {
  uint32_t struct_size; /* sizeof(ffpa_varlen_fwd_params), checked */
  uint32_t abi_version; /* FFPA_ATTN_ABI_VERSION                    */

  const void* q;
  const void* k;
  const void* v;
  void* o;
  float* lse; /* optional */
  const int32_t* cu_seqlens_q;  /* device, [batch + 1] */
  const int32_t* cu_seqlens_kv; /* device, [batch + 1] */
  const int32_t* seqused_kv;    /* optional, device, [batch]: sequence i uses only its first seqused_kv[i] key rows (clamped to its range) — a KV cache of
                                   fixed capacity per sequence (cu_seqlens_kv = multiples of the capacity) whose valid lengths live on the device and change
                                   between HIP-graph replays; FlashAttention's seqused_k / cache_seqlens.  NULL = every key row of the range */

  int32_t batch;          /* sequences */
  int32_t heads_q;        /* Hq  */
  int32_t heads_kv;       /* Hkv */
  int32_t head_dim;       /* D: any multiple of 8 in [8, 1024] */
  int32_t max_seqlen_q;   /* >= the longest query sequence */
  int32_t max_seqlen_kv;  /* >= the longest key sequence (launch-side hint) */

  int64_t q_stride[2]; /* elements: row, head */
  int64_t k_stride[2];
  int64_t v_stride[2];
  int64_t o_stride[2];
  int64_t lse_stride_head; /* elements between two heads of lse (>= total_q); ignored when lse == NULL */

  int32_t dtype;  /* enum ffpa_dtype */
  int32_t causal; /* 0 / 1: tail-aligned per sequence */

  float softmax_scale;     /* > 0 or < 0 or 0: as ffpa_fwd_params */
  float rescale_threshold; /* as ffpa_fwd_params: < 0 => 8.0 */
  uint32_t flags;          /* FFPA_FLAG_NO_XCD_REMAP, FFPA_FLAG_L2_PREFETCH / _NO_L2_PREFETCH, FFPA_FLAG_XCD_GROUP(), FFPA_FLAG_NO_PACK_GQA, FFPA_FLAG_KV_STREAM / _NO_KV_STREAM, FFPA_FLAG_DETERMINISTIC, FFPA_FLAG_FORCE_SPLITS; others ignored */
  uint32_t reserved;       /* 0 */

  /* KV SPLITS (ABI 6): batches whose (sequence, head) pairs leave most of the chip idle — a decode batch of a few long sequences: 8 sequences x 8 KV heads
   * are 64 workgroups for 256 CUs — split every sequence's KV range over num_splits workgroups (each sequence by ITS OWN length, read on the device:
   * ceil(tiles_i / splits) KV tiles per range), which write normalised fp32 partials + LSE to the workspace; a second kernel of the same call merges them
   * (the reference's decode stage 2, csrc/cuffpa/native/sm_80/split_kv.cuh:329-455).  Still nothing read on the host, still graph-capturable (two nodes).
   * Only launches with ONE row tile per (sequence, head) split (max_seqlen_q <= block rows: decode, speculative decoding, chunked prefill), and only with a
   * workspace: size from ffpa_attn_varlen_fwd_workspace_bytes().  A split launch's bits equal the unsplit launch's to fp32-merge rounding, not to the bit:
   * FFPA_FLAG_DETERMINISTIC or num_splits = 1 keeps one range per sequence. */
  void* workspace;          /* device scratch, 16-byte aligned; NULL => no split */
  uint64_t workspace_bytes;
  int32_t num_splits;       /* 0 = library heuristic, 1 = never split, n = at most n (with FFPA_FLAG_FORCE_SPLITS: exactly n where the key length allows) */
  int32_t total_q;          /* rows of q / o (>= cu_seqlens_q[batch]); 0 => no split (the partials are laid out [split, Hq, total_q, D]; a sequence whose rows
                               lie past total_q is skipped by a split launch — never stored outside the scratch) */
} ffpa_varlen_fwd_params;

/* Launch the packed-sequence forward on `stream` of the CURRENT device.  Asynchronous; returns an ffpa_status. */
int ffpa_attn_varlen_fwd(const ffpa_varlen_fwd_params* params, void* stream);

/* Scratch bytes the packed call wants for the KV split count its heuristic would pick with unlimited scratch (0 = it would not split). */
size_t ffpa_attn_varlen_fwd_workspace_bytes(const ffpa_varlen_fwd_params* params);

/* Its launch plan: out[0] = row tiles per (sequence, head) in the grid, out[1] = query rows per workgroup, out[2] = keys per tile,
 * out[3] = workgroups of the launch (all KV ranges), out[4] = KV ranges per sequence given params->workspace_bytes (ABI 6).  Returns an ffpa_status. */
int ffpa_attn_varlen_fwd_plan(const ffpa_varlen_fwd_params* params, int out[5]);

/* The kernel it runs, as text ("ffpa_fwd_m16_varlen_kernel<bf16, 512>").  Returns an ffpa_status. */
int ffpa_attn_varlen_fwd_kernel(const ffpa_varlen_fwd_params* params, char* buf, size_t n);

/* Capability / build queries.  Replaces the module attributes
 * CUDA_FWD_AVAILABLE, F16_ACC_AVAILABLE, ... (csrc/cuffpa/ffpa_api.cc:283-305). */
enum ffpa_query {
  FFPA_QUERY_ABI_VERSION = 0,
  FFPA_QUERY_FWD_AVAILABLE = 1,   /* 1 if the gfx950 kernels are in this build */
  FFPA_QUERY_MIN_HEAD_DIM = 2,    /* 8    */
  FFPA_QUERY_MAX_HEAD_DIM = 3,    /* 1024 */
  FFPA_QUERY_HEAD_DIM_MULTIPLE = 4, /* 8: kernels are built per multiple of 64; a head dim in between runs on the next one with
                                       the missing columns read as zeros in-kernel (no padded copies) */
  FFPA_QUERY_FP16_AVAILABLE = 5,
  FFPA_QUERY_DROPOUT_AVAILABLE = 6,
  FFPA_QUERY_DEBUG_KERNELS = 7,   /* 1 if FFPA_FLAG_DEBUG_SAFE_PATH kernels are built: only in the test-only twin library
                                     libffpa_attn_hip_test.so, never in the product library */
  /* what the launch plan's pricing reads from the CURRENT device (MI355X figures without one): the split rules are priced per device,
     not per SKU constant */
  FFPA_QUERY_DEVICE_CUS = 8,        /* compute units */
  FFPA_QUERY_DEVICE_CLOCK_MHZ = 9,  /* engine clock */
  FFPA_QUERY_DEVICE_HBM_GBPS = 10,  /* HBM peak (4 transfers x memory clock x bus width: HBM3 / HBM3E) */
  FFPA_QUERY_VARLEN_AVAILABLE = 11  /* 1 if ffpa_attn_varlen_fwd's kernels are in this build */
};
int ffpa_attn_query(int what);

/* Tile geometry the kernel uses for a head dim (for benches / roofline maths).
 * Returns 0 and fills rows-per-workgroup / keys-per-tile / dynamic LDS bytes,
 * or FFPA_ERR_BAD_HEADDIM. */
int ffpa_attn_fwd_tile_config(int head_dim, int* block_rows, int* block_keys,
                              int* lds_bytes);

/* Human-readable text for the last non-zero status on this thread. */
const char* ffpa_attn_last_error(void);

/* "ffpa-attn-amd <semver> gfx950" */
const char* ffpa_attn_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FFPA_ATTN_H_ */
