// ffpa_capi.hip — the extern "C" boundary declared in include/ffpa_attn.h.
//
// Host-side argument validation mirrors what the reference checks in its launcher
// (csrc/cuffpa/launch.cuh:79-129: shapes, bias layout; ffpa_api.cc:193-197: dtype)
// but reports through status codes instead of TORCH_CHECK, allocates nothing and
// keeps no global mutable state beyond write-once per-device caches (the reference's process-global backend hint,
// csrc/cuffpa/backend.h:16-27, has no equivalent here: every choice is per call).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "ffpa_attn.h"
#include "ffpa_fwd_kernel.h"
#include "ffpa_fwd_m16_kernel.h"  // (FFPA_M16_MIN_D: the head dims whose prefill launches run the 16x16x32 build)
#include "ffpa_varlen_merge.h"   // (stage 2 of a KV-split packed-sequence launch)
#include "ffpa_launch.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

typedef int (*launch_fn)(int, int, int, const ffpa::FwdArgs&, hipStream_t);
typedef void (*config_fn)(int, int*, int*, int*);

struct DimEntry {
  int d;
  launch_fn launch;
  config_fn config;
};

const DimEntry kDims[] = {
#define FFPA_ROW(D) {D, &ffpa::launch_fwd_d##D, &ffpa::tile_config_d##D},
    FFPA_FOR_EACH_HEAD_DIM(FFPA_ROW)
#undef FFPA_ROW
};

// the packed-sequence kernel: one launcher per head dim of the 16x16x32 build (ffpa_varlen_inst.hip)
typedef int (*varlen_fn)(int, int, const ffpa::FwdArgs&, const ffpa::VarlenArgs&, hipStream_t);
struct VarlenEntry {
  int d;
  varlen_fn launch;
};
const VarlenEntry kVarlenDims[] = {
#define FFPA_ROW(D) {D, &ffpa::launch_varlen_d##D},
    FFPA_FOR_EACH_VARLEN_HEAD_DIM(FFPA_ROW)
#undef FFPA_ROW
};

// Head dims are instantiated in multiples of 64; any multiple of 8 up to 1024 runs on the next instantiation with the
// columns past the caller's head dim read as zeros and never stored (the reference pads to its compiled multiples on
// the host, csrc/cuffpa/ffpa_api.cc:123-161).
int kernel_head_dim(int d) { return (d + 63) / 64 * 64; }

const DimEntry* find_dim(int d) {
  if (d <= 0 || d % 8 != 0) return nullptr;
  const int dk = kernel_head_dim(d);
  for (const DimEntry& e : kDims)
    if (e.d == dk) return &e;
  return nullptr;
}

// Launch plan: which tile variant, and over how many workgroups the KV axis is split.
struct Plan {
  int variant, br, bc, lds, nqt, nt, splits, tiles_per_split;
  int btile;  // 1: a bias with a row axis goes through LDS tiles staged one KV step ahead
  int m16;    // 1: the launch runs ffpa_fwd_m16_kernel (prefill tiles, head dim >= FFPA_M16_MIN_D), 0: ffpa_fwd_split_d_kernel
  int wide;   // 1: ... its wide-row tile, ffpa_fwd_m16w_kernel (launch variant 3)
  int mk;     // m16: the mask kind of the build (0 none, 1 additive bias, 2 boolean mask / ranges)
  int bias_raw;  // FwdArgs.bias_cache_raw
  int bias_lds;  // FwdArgs.bias_lds: > 0 bytes of the key-bias row cache, < 0 -(bytes of the bias-tile staging areas), 0 neither
  int pair;      // 1: row tiles i and nqt - 1 - i share a workgroup (FwdArgs.pair_tiles): the grid holds (nqt + 1) / 2 workgroups per (batch, head)
  int chunk;     // > 1: a causal GQA launch in the HEAD-CHUNK order — (head chunk, batch, row tile, head in chunk), chunks of this many heads of one KV group: the same row
                 //      tile of the group's heads runs at the same time on the same keys (ffpa_fwd_m16_varlen_kernel in its dense mode: the packed-sequence kernel's order)
  int tile_ranges;  // 1: a causal launch whose KV splits are PER ROW TILE — every row tile shares out the KV tiles up to its own diagonal (the same kernel's dense mode)
  size_t ws_bytes;
};

// Per-device facts, looked up once per device and process (write-once caches: the only state the library keeps;
// std::atomic so that concurrent first calls from several host threads are race-free).
constexpr int kMaxDevices = 64;
std::atomic<int> g_cu_count[kMaxDevices];  // 0 = not looked up yet
std::atomic<int> g_arch_ok[kMaxDevices];   // 0 = not looked up yet, 1 = gfx950, 2 = something else

// FFPA_HIP_FAKE_CUS=<n> makes the launch plan price a device of n compute units — a SUPPORTED override: a process that owns a CU-masked
// slice of the GPU (HSA_CU_MASK, partition modes) still reads the whole chip's count from the attribute below; the plan fuzz of tests/test_fwd_gpu.py and
// the CPU plan tables of tests/test_capi.py walk 128 / 256 / 304 with it.  Read on every call while set (no cached state to go stale); the kernels never
// see it — any grid is correct on any device, the plan only decides how fast.
int device_cu_count() {
  if (const char* fake = getenv("FFPA_HIP_FAKE_CUS")) {
    const int n = atoi(fake);
    if (n > 0) return n;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
  int n = g_cu_count[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    g_cu_count[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

// What the split pricing needs to know about the device besides its CU count: the matrix rate of ONE compute unit and the HBM bandwidth — both read
// from the device (engine clock; memory clock x bus width) instead of being MI355X constants, scaled by what this library's kernels were MEASURED to
// sustain of them on gfx950 (PlanTunables below: the same fractions the bench lines report as roofline.frac).  A part with other
// clocks or another stack count prices itself; without a device (plan queries on a CPU box) the MI355X figures are the fallback.
struct DeviceRates {
  double cu_flops;   // dense bf16 MFMA peak of one CU: 4 SIMDs x 1024 FLOP / clk x engine clock
  double hbm_bytes;  // HBM peak, bytes / s
};
// (one word per device — engine clock in kHz << 32 | HBM bandwidth in MB / s —, so that a concurrent first caller sees both figures or neither)
std::atomic<unsigned long long> g_rates[kMaxDevices];  // 0 = not looked up yet

DeviceRates device_rates() {
  DeviceRates r = {4.0 * 1024.0 * 2.4e9, 8.0e12};  // MI355X: 2.4 GHz, 8 TB/s (MI355X_MICROARCH.md)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    (void)hipGetLastError();
    return r;
  }
  unsigned long long packed = g_rates[dev].load(std::memory_order_acquire);
  if (packed == 0) {
    int clk = 0, mclk = 0, bus = 0;
    long long mbps = 0;
    if (hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, dev) != hipSuccess || clk <= 0) clk = 2400000;
    // HBM3 / HBM3E (every gfx94x / gfx950 part; this library only loads on gfx950): four transfers per reported memory clock x bus width —
    // MI355X: 2000 MHz x 8192 bit x 4 / 8 = 8.19 TB/s, MI300X: 1300 MHz -> 5.3 TB/s.  A driver that does not report them leaves the MI355X figure
    if (hipDeviceGetAttribute(&mclk, hipDeviceAttributeMemoryClockRate, dev) == hipSuccess && mclk > 0 &&
        hipDeviceGetAttribute(&bus, hipDeviceAttributeMemoryBusWidth, dev) == hipSuccess && bus > 0)
      mbps = (long long)(4.0 * (double)mclk * 1e3 * (double)bus / 8.0 / 1e6);
    if (mbps < 500000 || mbps > 20000000) mbps = 8000000;  // (0.5 ... 20 TB/s: anything else is a driver reporting another unit)
    (void)hipGetLastError();
    packed = ((unsigned long long)(unsigned)clk << 32) | (unsigned long long)(unsigned)mbps;
    g_rates[dev].store(packed, std::memory_order_release);
  }
  r.cu_flops = 4.0 * 1024.0 * (double)(packed >> 32) * 1e3;
  r.hbm_bytes = (double)(packed & 0xffffffffull) * 1e6;
  return r;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// The launch plan.  Every FITTED number it uses lives in this one block, keyed by architecture (this library holds gfx950 code objects only:
// one entry); the rules that use them are a table further down, each with its predicate, its candidates, its margin and the file under
// profiles/ that justifies it.
// ------------------------------------------------------------------------------------------------------------------------------------
struct PlanTunables {
  // fraction of a CU's MFMA peak one workgroup of the prefill tiles sustains while it walks KV tiles (D <= 512 / split-D tiles), and the fraction of
  // the HBM peak the partial-write + merge traffic of a KV-split launch moves at (5.0 / 4.0 TFLOP/s per CU, 5 TB/s on MI355X: profiles/r04_launch_side.txt)
  double tile_rate_frac_d512, tile_rate_frac_splitd, merge_bw_frac;
  double fixed_tiles;          // per-workgroup fixed cost (prologue, epilogue, dispatch) in KV-tile times (profiles/r04_launch_side.txt)
  double max_auto_ws_bytes;    // the pricing never asks a caller for more scratch than this on its own (a forced num_splits may)
  double wide_gain;            // rows per unit time, wide-row tile : 32-row tile — the low end of the measured 2 ... 7 % (profiles/r05_wide_tile.txt)
  int wide_min_wg_per_cu;      // ragged launches (causal flag, masks, ranges): workgroups per CU from which the wide tile is taken (profiles/r05_wide_tile.txt)
  int min_tiles_prefill, min_tiles_short;  // KV tiles a split range holds at least: the merge stays cheap (profiles/r03_decode_splits.txt, r04_launch_side.txt)
  int short_one_per_cu_min_d, short_one_per_cu_lds;  // short-query tiles that want ONE workgroup per CU: head dims from here up, or tiles above this much LDS (profiles/r03_decode_splits.txt)
  int det_tiles_per_split;     // FFPA_FLAG_DETERMINISTIC: KV tiles per split range of a short-query launch (a function of the KV length alone)
  int pair_max_row_tiles;      // causal launches pair row tiles i and n - 1 - i in one workgroup up to this many row tiles per head (profiles/r06_pair_tiles.txt)
  // KV splits of the packed-sequence call (varlen_plan; profiles/r06_varlen_splits.txt): a batch of several sequences is ragged, and ranges beyond "one workgroup per
  // slot" even out what the longest sequence would otherwise run alone — up to this many workgroups per slot, at least this many KV tiles of the longest sequence per
  // range, and partials (written + read back) of at most this fraction of the K + V bytes the launch side can estimate
  int varlen_balance_wgs, varlen_balance_min_tiles;
  double varlen_partial_frac;
  // per-row-tile KV ranges of dense causal launches of one round (pick_tile_ranges; profiles/r06_tile_ranges.txt): keys a range of the average row tile keeps at
  // least (head dims <= 512 / the split-D tiles), and the head dim below which the launch takes three ranges instead of two (two workgroups fit a CU there)
  double tile_ranges_min_keys, tile_ranges_min_keys_splitd;
  int tile_ranges_three_below_d;
};
constexpr PlanTunables kPlanGfx950 = {
    5.0e12 / 9.8304e12, 4.0e12 / 9.8304e12, 5.0e12 / 8.0e12,
    4.0,
    1024.0 * 1048576.0,
    1.04,
    8,
    8, 4,
    320, 80 * 1024,
    16,
    32,
    4, 16,
    0.02,
    640.0, 2048.0,
    256,
};
constexpr const PlanTunables& kT = kPlanGfx950;

// FFPA_OK iff the current device is a gfx950 (the only target of the embedded code objects)
int check_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return fail(FFPA_ERR_NO_DEVICE, "no current HIP device");
  }
  int ok = (dev >= 0 && dev < kMaxDevices) ? g_arch_ok[dev].load(std::memory_order_relaxed) : 0;
  if (ok == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      (void)hipGetLastError();
      return fail(FFPA_ERR_NO_DEVICE, "hipGetDeviceProperties(%d) failed", dev);
    }
    ok = strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 2;
    if (dev >= 0 && dev < kMaxDevices) g_arch_ok[dev].store(ok, std::memory_order_relaxed);
    if (ok == 2) return fail(FFPA_ERR_NO_DEVICE, "device %d is %s: this library holds gfx950 (MI355X) kernels only", dev, prop.gcnArchName);
  }
  if (ok == 2) return fail(FFPA_ERR_NO_DEVICE, "device %d is not a gfx950 (MI355X)", dev);
  return FFPA_OK;
}

// What every rule of the plan looks at: the call, the tile the launch would run, the device, and the pricing model's inputs.
struct PlanCtx {
  const ffpa_fwd_params* p;
  const DimEntry* de;
  Plan* pl;
  int64_t cus, base;     // compute units; workgroups of the unsplit launch (batch x heads x row tiles)
  int dk;                // head dim of the kernel instantiation
  double nt_eff;         // KV tiles an average row tile walks (causal flag: up to its diagonal)
  int nt_visible;        // KV tiles any row can see: a split range past them would be workgroups that do nothing
  double tile_s, out_elems, merge_bw;
  bool priced;           // the split count is the pricing's to choose (prefill tile, no forced count, scratch offered)
  bool deterministic;    // FFPA_FLAG_DETERMINISTIC
};

// The pricing model of a prefill launch split over s KV ranges (round 4, profiles/r04_launch_side.txt; it over-prices the splits by ~ 10 % on every measured shape):
//   time(s) = rounds(s) x (tiles per split + fixed_tiles of per-workgroup fixed cost) x tile time  +  (8 s + 2) bytes per output element / merge bandwidth
double predicted(const PlanCtx& c, int64_t s) {
  const double rounds = (double)((c.base * s + c.cus - 1) / c.cus);
  return rounds * (c.nt_eff / s + kT.fixed_tiles) * c.tile_s + (s > 1 ? c.out_elems * (8.0 * s + 2.0) / c.merge_bw : 0.0);
}
// a split count is admissible for the pricing's own rules if its last KV range still holds keys some row can see and its scratch stays under the cap
bool admissible(const PlanCtx& c, int64_t s) {
  const int64_t tps = (c.pl->nt + s - 1) / s;
  return (s - 1) * tps < c.nt_visible && (double)s * c.out_elems / c.dk * (c.dk + 1.0) * 4.0 <= kT.max_auto_ws_bytes;
}

// A PRICED rule: where `applies`, every split count of `range` (ascending; the walk stops at the first count that leaves a range fewer than
// min_tiles_prefill KV tiles) is priced against the count the plan holds so far, and the cheapest one that beats `margin` x that price is taken.
struct PricedRule {
  const char* name;
  bool (*applies)(const PlanCtx&, int64_t cur);
  void (*range)(const PlanCtx&, int64_t cur, int64_t* lo, int64_t* hi);
  double margin;
  const char* evidence;
};
int64_t run_priced(const PlanCtx& c, const PricedRule& r, int64_t cur) {
  if (!r.applies(c, cur)) return cur;
  int64_t lo = 0, hi = -1, pick = cur;
  r.range(c, cur, &lo, &hi);
  const double t0 = predicted(c, cur);
  double best = t0;
  for (int64_t s = lo; s <= hi; ++s) {
    if (c.pl->nt / s < kT.min_tiles_prefill) break;
    if (!admissible(c, s)) continue;
    const double t = predicted(c, s);
    if (t < r.margin * t0 && t < best) best = t, pick = s;
  }
  return pick;
}
const PricedRule kRaggedRound = {
    // a prefill launch of a little over one round of workgroups — 288 ... 384 on 256 CUs — takes two rounds' time.  Split over 2 or 3 KV ranges it fills whole
    // rounds (9 heads x 32 row tiles x 3 = 864 = 3.4 rounds), and on a long context the partials and their merge cost little next to that: B1 H9 / H10 / H11 /
    // H12 x Nq 4096 x Nkv 8192 D512 + 18 / + 15 / + 10 / + 5 %, H40 x Nq 1024 + 14 %, D = 1024 H5 + 19 %; against 2048 keys or under the causal flag the same
    // splits LOSE 18 ... 37 % — the rule prices both sides and splits only for a predicted gain of 10 %.  The same pricing serves a launch of PART of a round
    // (CUs / 2 < workgroups < CUs): 160 workgroups in three KV ranges are 480 = two rounds of a third of the length — B1 H5 x Nq 4096 D512 + 10 % at 8192 keys,
    // + 19 % at 16384, D = 320 + 7 %, H20 x Nq 1024 + 8 %; 192 and 224 workgroups (H6, H7, D = 1024 H3) gain from no split count, and the model picks none.
    "ragged round",
    [](const PlanCtx& c, int64_t) {
      return c.priced && 2 * c.base > c.cus && 2 * c.base <= 3 * c.cus && c.p->bias == nullptr && c.p->kv_bounds == nullptr && !(c.p->dropout_p > 0.f);
    },
    [](const PlanCtx&, int64_t, int64_t* lo, int64_t* hi) { *lo = 2, *hi = 3; },
    0.9, "profiles/r04_launch_side.txt"};
const PricedRule kTwoRounds = {
    // under-filled launches whose CUs / workgroups is far from a whole number (96 workgroups: two ranges fill 192 of 256 CUs): a count that makes two rounds of
    // shorter workgroups can beat the one-round count on a long context — 96 workgroups x 5 = 480: + 6 ... 11 % at 16384 keys, + 7 % at D = 1024 / 8192 keys,
    // + 2 % at D = 512 / 8192 keys; 80 x 3 and 56 x 4 stay.  Both sides pay partials and a merge, so the margin is 5 % here.
    "two rounds of shorter workgroups",
    [](const PlanCtx& c, int64_t cur) { return c.priced && c.base * 2 <= c.cus && cur >= 2; },
    [](const PlanCtx&, int64_t cur, int64_t* lo, int64_t* hi) { *lo = cur + 1, *hi = 3 * cur < ffpa::kMergeMaxSplits ? 3 * cur : ffpa::kMergeMaxSplits; },
    0.95, "profiles/r04_launch_side.txt"};

// The wide-row prefill tile (ffpa_fwd_m16w_kernel.h: 16 RH rows per wave, RH sized to the accumulator file; 64-key tiles, double-buffered, one barrier
// per KV step): does this launch take it?  The builds that exist: no additive bias, no dropout.  Measured on D = 320 (profiles/r05_wide_tile.txt, interleaved
// same-box A/B): a row of the 192-row tile costs 2 ... 7 % less than a row of the 128-row tile, so what decides is how the launch quantises into rounds of
// workgroups — B2 Hq32 x Nq 8192 (config 4 without its mask): 2752 workgroups = 11 rounds of 192 rows against 16 of 128: + 2.3 %; B1 H32 x Nq 8192: 1376 =
// 5.4 -> SIX rounds against 8: - 4 %.  Launches whose workgroups differ in length (causal flag, masks, mask ranges: longest first, the last round's tail is short
// workgroups) have no such quantum but need enough workgroups per CU for the lengths to even out: config 4 (10.75 per CU) + 1.0 ... 1.6 %, B1 H32 causal 8192
// (5.4 per CU) - 3 %.
bool pick_wide_tile(const ffpa_fwd_params* p, const DimEntry* de, const Plan& pl, int64_t cus) {
  if (pl.variant != 0 || (p->flags & (FFPA_FLAG_DEBUG_SAFE_PATH | FFPA_FLAG_NO_WIDE_TILE | FFPA_FLAG_DETERMINISTIC))) return false;
  if (p->dropout_p > 0.f || !(p->bias == nullptr || p->bias_dtype == FFPA_BIAS_BOOL8)) return false;
  int br = 0, bc = 0, lds = 0;
  de->config(3, &br, &bc, &lds);
  if (br <= 0) return false;
  if (p->flags & FFPA_FLAG_WIDE_TILE) return true;
  const int64_t wgs_wide = (int64_t)p->batch * p->heads_q * ((p->seqlen_q + br - 1) / br);
  const int64_t wgs_now = (int64_t)p->batch * p->heads_q * ((p->seqlen_q + pl.br - 1) / pl.br);
  if (p->causal || p->bias != nullptr || p->kv_bounds != nullptr) return wgs_wide >= kT.wide_min_wg_per_cu * cus;
  if (wgs_wide < 2 * cus) return false;  // (launches of a round or two: the KV-split rules were measured on the 128-row tile)
  const double t_wide = (double)((wgs_wide + cus - 1) / cus) * br / kT.wide_gain, t_now = (double)((wgs_now + cus - 1) / cus) * pl.br;
  return t_wide < t_now;
}

// The pricing's inputs for this call (`pl` holds variant, tile and tile counts already).
PlanCtx plan_context(const ffpa_fwd_params* p, const DimEntry* de, Plan* pl, int64_t cus) {
  PlanCtx c = {};
  c.p = p, c.de = de, c.pl = pl, c.cus = cus;
  c.base = (int64_t)p->batch * p->heads_q * pl->nqt;
  c.dk = kernel_head_dim(p->head_dim);
  // KV tiles an average row tile walks: all of them — or, under the causal flag, those up to its diagonal: rows r see keys <= r + causal_offset, the mean
  // over the rows is causal_offset + Nq / 2, clamped to [0, Nkv] (top-left causal against a long context: Nq / 2 keys, not Nkv / 2; tail-aligned: Nkv - Nq / 2)
  c.nt_eff = (double)pl->nt;
  int64_t visible_end = p->seqlen_kv;  // keys at and past this index are hidden from EVERY row of the launch
  if (p->causal) {
    double mean = (double)p->causal_offset + 0.5 * (double)(p->causal_row_mod ? p->causal_row_mod : p->seqlen_q);
    mean = mean < 0.0 ? 0.0 : (mean > (double)p->seqlen_kv ? (double)p->seqlen_kv : mean);
    c.nt_eff = mean / pl->bc > 1.0 ? mean / pl->bc : 1.0;
    const int64_t last = (int64_t)(p->causal_row_mod ? p->causal_row_mod : p->seqlen_q) - 1 + p->causal_offset + 1;
    visible_end = last < 0 ? 0 : (last < p->seqlen_kv ? last : p->seqlen_kv);
  }
  c.nt_visible = (int)((visible_end + pl->bc - 1) / pl->bc);
  const DeviceRates rates = device_rates();
  c.tile_s = 4.0 * pl->br * pl->bc * c.dk / ((c.dk > 512 ? kT.tile_rate_frac_splitd : kT.tile_rate_frac_d512) * rates.cu_flops);  // one KV tile of one workgroup
  c.out_elems = (double)p->batch * p->heads_q * p->seqlen_q * c.dk;
  c.merge_bw = kT.merge_bw_frac * rates.hbm_bytes;
  c.deterministic = (p->flags & FFPA_FLAG_DETERMINISTIC) != 0;
  c.priced = pl->variant == 0 && !(p->flags & (FFPA_FLAG_DEBUG_SAFE_PATH | FFPA_FLAG_FORCE_SPLITS | FFPA_FLAG_DETERMINISTIC)) && p->num_splits == 0 && p->workspace != nullptr;
  return c;
}

// KV-split launches (partials + LSE merge): over how many KV ranges?  Short-query tiles: occupancy heuristic cf. select_decode_num_splits
// (native/launch.cuh:17-67) — aim at one (head dims >= 320) or two workgroups per CU, at least 4 KV tiles per split so the merge stays cheap.  Prefill tiles
// split when the launch would leave more than half of the chip idle (chunked prefill against a long context with few heads per GPU: one workgroup per CU, at
// least 8 KV tiles per split), or when a priced rule says so.
int64_t pick_splits(const PlanCtx& c) {  // (0: this launch does not split — no clamp applies)
  const ffpa_fwd_params* p = c.p;
  const Plan& pl = *c.pl;
  if (c.deterministic) {
    // batch-invariant bits: a prefill launch never splits; a short-query launch splits by the KV length ALONE (a fixed number of tiles per range) — the
    // plan of a (batch, head) slice then does not depend on how many slices share the launch, and neither does a single bit of its output
    if (pl.variant == 0 || p->num_splits == 1) return 0;
    return (pl.nt + kT.det_tiles_per_split - 1) / kT.det_tiles_per_split;
  }
  const bool forced = pl.variant == 0 && (p->flags & FFPA_FLAG_FORCE_SPLITS) && p->num_splits > 1;  // (the flag: sweeps of the rules, tools/gpu_prefill_splits.py)
  const int64_t ragged = run_priced(c, kRaggedRound, 1);
  const bool underfilled = pl.variant == 0 && !(p->flags & FFPA_FLAG_DEBUG_SAFE_PATH) && (c.base * 2 <= c.cus || ragged > 1 || forced);
  if (!((pl.variant == 1 || underfilled) && p->num_splits != 1)) return 0;
  // short-query tiles, measured (tools/gpu_decode_splits.py, profiles/r03_decode_splits.txt): head dims >= 320 want ONE workgroup per CU — their tiles are
  // 20 KiB and up, one workgroup keeps enough bytes in flight, and half the splits are half the partials to write and merge (- 3 ... 14 % per step
  // against two per CU; rounded DOWN: at most two of these workgroups fit a CU, above D = 512 one, and a launch a little over one per CU takes twice
  // as long as one a little under) — the small head dims two (8 KiB tiles at D = 128: one per CU is 40 ... 60 % slower, three or four 10 ... 25 %)
  const bool sq_one_per_cu = c.dk >= kT.short_one_per_cu_min_d || pl.lds > kT.short_one_per_cu_lds;  // (or tiles of which only one workgroup fits a CU)
  int64_t want = pl.variant == 1 ? (sq_one_per_cu ? c.cus / c.base : (2 * c.cus + c.base - 1) / c.base) : c.cus / c.base;
  if (forced) want = p->num_splits;
  if (ragged > 1) want = ragged;
  want = run_priced(c, kTwoRounds, want);
  return want < 1 ? 1 : want;
}

// What every split count is clamped by, whoever chose it: tiles per range, ranges nobody can see, the merge kernel's LDS, the caller's request and scratch.
int64_t clamp_splits(const PlanCtx& c, int64_t want) {
  const ffpa_fwd_params* p = c.p;
  const Plan& pl = *c.pl;
  const int min_tiles = pl.variant == 1 ? kT.min_tiles_short : kT.min_tiles_prefill;
  const int64_t cap = pl.nt / min_tiles > 0 ? pl.nt / min_tiles : 1;
  if (want > cap) want = cap;
  // prefill under the causal flag: no KV range entirely behind every row's diagonal (top-left causal against a long context: rows see at most Nq keys —
  // ranges past them would be workgroups that do nothing, their partials and the merge pure cost)
  if (pl.variant == 0 && !(p->flags & FFPA_FLAG_FORCE_SPLITS))
    while (want > 1 && (want - 1) * ((pl.nt + want - 1) / want) >= c.nt_visible) --want;
  if (want > ffpa::kMergeMaxSplits) want = ffpa::kMergeMaxSplits;  // the merge kernel keeps the split weights in LDS
  if (p->num_splits > 1 && want > p->num_splits) want = p->num_splits;
  if (want < 1) want = 1;
  const size_t per_split = (size_t)p->batch * p->heads_q * p->seqlen_q * ((size_t)c.dk + 1) * sizeof(float);
  if (p->workspace == nullptr) want = 1;
  else if ((uint64_t)want * per_split > p->workspace_bytes) want = (int64_t)(p->workspace_bytes / per_split);
  return want < 1 ? 1 : want;
}

// The 16x16x32 prefill build of this call (ffpa_fwd_inst.hip): no bias / boolean masks and mask ranges / additive biases (and anything next to dropout) — and,
// for an additive bias, where the initial S^T accumulators come from (an LDS row cache, a ring, staged tiles, or element loads).
void pick_m16_build(const ffpa_fwd_params* p, const DimEntry* de, Plan& pl, int dk) {
  const bool no_bias = p->bias == nullptr && p->kv_bounds == nullptr;
  const bool additive = p->bias != nullptr && p->bias_dtype >= FFPA_BIAS_FP16 && p->bias_dtype <= FFPA_BIAS_FP32;
  pl.mk = no_bias ? 0 : ((additive || p->dropout_p > 0.f) ? 1 : 2);
  if (pl.mk != 1) return;
  // 64-key tiles at every head dim <= 512 (the LDS also holds the bias)
  de->config(2, &pl.br, &pl.bc, &pl.lds);
  pl.nt = (p->seqlen_kv + pl.bc - 1) / pl.bc;
  const int esz = p->bias_dtype == FFPA_BIAS_FP32 ? 4 : 2;
  const int stagers = dk > 512 ? 2 : 4;  // waves that carry a bias tile (D > 512: one of the two waves of a row block)
  const bool lds_ok = additive && pl.splits == 1 && !(p->flags & FFPA_FLAG_NO_BIAS_LDS);
  const int64_t cache = (int64_t)pl.nt * pl.bc * 4;  // a key bias as fp32 / scale, a whole number of tiles
  const int64_t stage = (int64_t)stagers * 32 * pl.bc * esz;
  bool stage_ok = lds_ok && p->bias_stride[3] == 1 && reinterpret_cast<uintptr_t>(p->bias) % 16 == 0 && p->bias_stride[2] < (1LL << 24) &&
                  pl.lds + stage <= 160 * 1024;
  for (int i = 0; i < 3 && stage_ok; ++i) stage_ok = (p->bias_stride[i] * esz) % 16 == 0;
  const bool cache16 = esz == 2 && pl.lds + cache / 2 <= 160 * 1024;  // ... or as the caller's 16-bit elements, converted per step
  // split-D tiles with the softmax pipeline (D > 512, D % 128 == 0), nothing but a key bias: the key-bias build runs the pipeline too (round 5); its LDS is the
  // unmasked build's (6 KiB of exchange per wave) + a RING of fp32 bias entries — a power of two, at least 2048 (all D = 1024 has left), refilled half a ring
  // at a time 1024+ keys ahead of the walk (ffpa_fwd_m16_kernel.h)
  const bool ring = lds_ok && p->bias_stride[2] == 0 && dk > 512 && dk % 128 == 0 && p->kv_bounds == nullptr && !(p->dropout_p > 0.f);
  if (ring) {
    int br0 = 0, bc0 = 0, lds0 = 0;
    de->config(4, &br0, &bc0, &lds0);
    int bytes = 8192;
    while (2 * bytes <= 160 * 1024 - lds0 && bytes < 65536) bytes *= 2;
    pl.lds = lds0;
    pl.bias_raw = 0;
    pl.bias_lds = bytes;
    pl.mk = 3;
  } else if (lds_ok && p->bias_stride[2] == 0 && (pl.lds + cache <= 160 * 1024 || cache16)) {
    pl.bias_raw = pl.lds + cache <= 160 * 1024 ? 0 : 1;
    pl.bias_lds = (int)(pl.bias_raw ? cache / 2 : cache);
    if (p->kv_bounds == nullptr && !(p->dropout_p > 0.f)) pl.mk = 3;  // nothing but a cached key bias: the lean key-bias build
  } else if (stage_ok) {
    pl.btile = 1;
    pl.bias_lds = -(int)stage;
  }
}

// 32x32x16 build (head dims <= 64): a 16-bit bias with a real row axis is staged through LDS by LDS-DMA one KV step ahead
// (4 waves x [32 rows x 64 keys]); those launches run the build with 64-key tiles at every head dim (tile config variant 2)
void pick_small_d_bias_tiles(const ffpa_fwd_params* p, const DimEntry* de, Plan& pl) {
  if (pl.m16 || pl.variant != 0 || pl.splits != 1 || p->bias == nullptr || p->bias_stride[2] == 0 || p->bias_stride[3] != 1) return;
  if (!(p->bias_dtype == FFPA_BIAS_FP16 || p->bias_dtype == FFPA_BIAS_BF16) || (p->flags & (FFPA_FLAG_NO_BIAS_LDS | FFPA_FLAG_DEBUG_SAFE_PATH)) || p->dropout_p > 0.f) return;
  bool ok = reinterpret_cast<uintptr_t>(p->bias) % 16 == 0 && p->bias_stride[2] < (1LL << 24);
  for (int i = 0; i < 3; ++i) ok = ok && (p->bias_stride[i] * 2) % 16 == 0;
  int br = 0, bc = 0, lds = 0;
  de->config(2, &br, &bc, &lds);
  if (!(ok && lds + 4 * 32 * bc * 2 <= 160 * 1024)) return;
  pl.btile = 1;
  pl.br = br, pl.bc = bc, pl.lds = lds;
  pl.nt = (p->seqlen_kv + pl.bc - 1) / pl.bc;
}

// Paired row tiles (ffpa_fwd_m16_kernel.h, FwdArgs::pair_tiles): under the causal flag workgroup i of a head walks row tile nqt - 1 - i and then row tile i — equal
// work per workgroup, half the dispatches, bit-identical outputs.  The builds that can: the 16x16x32 prefill kernel (not its wide-row tile), unsplit, no bias /
// ranges / packed rows.  Measured (interleaved same-box A/B over 19 causal shapes, profiles/r06_pair_tiles.txt): what it buys is the launch's TAIL — the last
// round of an unpaired launch is its short workgroups, most of a CU's time there is idle — so it pays where a head has FEW row tiles: 16 tiles (B4 H32 N 2048 D512)
// + 2 ... 12.7 % (three boxes), 32 tiles 0 ... + 3.6 % (D512) / + 2 ... 7.1 % (D320), 64 tiles - 1 ... + 2.4 % (D 128 ... 768; D = 1024 N 4096 + 0.3 %: not taken), 96 tiles - 0.9 %,
// 128 tiles - 1.2 % (D512 N 16384) ... - 7 / - 9 % (D = 1024 N 8192) — and only where the diagonal makes tiles unequal (tail-aligned causal against a context four
// times the query: - 3.1 %).  The rule: up to 32 row tiles per head.
bool pick_pair_tiles(const ffpa_fwd_params* p, const Plan& pl) {
  if (!pl.m16 || pl.wide || pl.splits != 1 || !p->causal || p->bias != nullptr || p->kv_bounds != nullptr || p->causal_row_mod != 0 || pl.nqt < 2) return false;
  if (p->flags & (FFPA_FLAG_NO_PAIR_TILES | FFPA_FLAG_DEBUG_SAFE_PATH)) return false;
  if (p->flags & FFPA_FLAG_PAIR_TILES) return true;
  // keys the first / the last row tile can see: the pairing evens out what the diagonal makes unequal — at least a factor of two between them
  auto clampkv = [&](int64_t x) { return x < 0 ? (int64_t)0 : (x > p->seqlen_kv ? (int64_t)p->seqlen_kv : x); };
  const int64_t lo = clampkv((int64_t)p->causal_offset + pl.br), hi = clampkv((int64_t)p->causal_offset + p->seqlen_q);
  return pl.nqt <= kT.pair_max_row_tiles && 2 * lo <= hi;
}

// Heads that walk a sequence side by side in the head-chunk order: the largest divisor of the KV group's size that leaves a multiple of eight chunks (every XCD
// then owns whole chunks of every batch element / sequence: balance); 1 = plain head-major order.
int pick_head_chunk(int heads_q, int group) {
  for (int c = group; c > 1; --c)
    if (group % c == 0 && heads_q % c == 0 && (heads_q / c) % 8 == 0) return c;
  return 1;
}

// Causal GQA launches in the head-chunk order (round 6: found on the packed-sequence kernel, whose order it is).  A head's row tiles alone fill an XCD for a round,
// so in the dense order the heads of a KV group stream the group's K / V one after the other; interleaved per row tile they stream it together.  Same tile, same
// bits; interleaved same-run A/B of the two orders on dense shapes (tools/gpu_dense_vs_packed.py, profiles/r06_head_chunks.txt): Hq 32 / Hkv 8 N 8192 causal D 512
// + 3.2 %, D 1024 + 4.3 %, D 320 + 2.8 %, B 2 N 4096 + 1.2 %, B 4 N 2048 + 1.8 % (over the paired-tile launch); non-causal +- 0; MHA - 3.7 % with chunks of 4 heads
// (four K / V streams at once instead of one) -> chunks never span KV groups.  Builds without bias / mask ranges / dropout; every row must see a key (the kernel's
// empty-row contract is the packed call's: 0, not SDPA's NaN).
int pick_dense_head_chunk(const ffpa_fwd_params* p, const Plan& pl, int dk) {
  if (!pl.m16 || pl.wide || pl.splits != 1 || pl.mk != 0 || !p->causal || p->causal_offset < 0 || p->causal_row_mod != 0 || p->dropout_p > 0.f || dk < FFPA_M16_MIN_D) return 1;
  if (p->flags & (FFPA_FLAG_NO_HEAD_CHUNKS | FFPA_FLAG_DEBUG_SAFE_PATH | FFPA_FLAG_PAIR_TILES)) return 1;
  if (p->lse != nullptr && (int64_t)p->batch * p->heads_q * p->seqlen_q >= (1LL << 31)) return 1;
  return pick_head_chunk(p->heads_q, p->heads_q / p->heads_kv);
}

// PER-ROW-TILE KV ranges for causal launches of ONE ROUND of workgroups (CUs / 2 < workgroups <= CUs: a whole prompt with a few heads per GPU — 8 heads x 4096
// tokens are 256 row tiles).  Longest first, the round ends on a few long row tiles while the short ones' CUs idle; uniform KV ranges (the dense kernel's) do not help —
// a short row tile sees nothing of the later ranges, the long ones still walk the first range whole: 2 ranges - 15 %, tools/gpu_prefill_splits.py c_* —, ranges of
// every row tile's OWN visible keys do (the packed-sequence kernel computes them on the device: its dense mode): 8 heads x 4096 x 4096 D 512 187 -> 165 us against
// 199 us of the paired-tile launch, D 128 85 -> 70 us (profiles/r06_tile_ranges.txt).  Taken when the longest row tile walks >= 1.5 x the average one and a range
// of the average one keeps enough keys (below).  The builds the dense mode has (pick_dense_head_chunk); fp32 partials + merge: equal to the one-range launch
// to rounding — FFPA_FLAG_DETERMINISTIC / num_splits = 1 / FFPA_FLAG_NO_TILE_RANGES keep one range.
int pick_tile_ranges(const ffpa_fwd_params* p, const Plan& pl, const PlanCtx& c) {
  if (!pl.m16 || pl.wide || pl.splits != 1 || pl.mk != 0 || !p->causal || p->causal_offset < 0 || p->causal_row_mod != 0 || p->dropout_p > 0.f || c.dk < FFPA_M16_MIN_D) return 0;
  if (p->flags & (FFPA_FLAG_NO_TILE_RANGES | FFPA_FLAG_DEBUG_SAFE_PATH | FFPA_FLAG_PAIR_TILES | FFPA_FLAG_DETERMINISTIC)) return 0;
  if (p->workspace == nullptr || p->num_splits == 1 || pl.nqt < 2) return 0;
  if ((int64_t)p->batch * p->heads_q * p->seqlen_q >= (1LL << 31)) return 0;
  int64_t want = 0;
  if ((p->flags & FFPA_FLAG_TILE_RANGES) && (p->flags & FFPA_FLAG_FORCE_SPLITS) && p->num_splits > 1) {
    want = p->num_splits < pl.nt ? p->num_splits : pl.nt;  // (sweeps and tests: exactly n)
  } else if (c.priced && 2 * c.base > c.cus && c.base <= c.cus && (double)c.nt_visible >= 1.5 * c.nt_eff) {
    // a range of the average row tile keeps at least tile_ranges_min_keys keys (the partials and their merge: measured neutral at 512 keys per range — 16 heads x
    // 2048 x 2048 - 1 % —, + 22 % at 750, + 15 ... 33 % at 1024; the split-D tiles - 4 % at 1024, + 8 % at 2048); small tiles (two workgroups per CU) take three
    const double avg_keys = c.nt_eff * pl.bc, min_keys = c.dk > 512 ? kT.tile_ranges_min_keys_splitd : kT.tile_ranges_min_keys;
    for (int64_t w = c.dk < kT.tile_ranges_three_below_d ? 3 : 2; w >= 2 && want == 0; --w)
      if (avg_keys / (double)w >= min_keys) want = w;
  }
  if (want > ffpa::kMergeMaxSplits) want = ffpa::kMergeMaxSplits;
  const size_t per_split = (size_t)p->batch * p->heads_q * p->seqlen_q * ((size_t)c.dk + 1) * sizeof(float);
  if ((uint64_t)want * per_split > p->workspace_bytes) want = (int64_t)(p->workspace_bytes / per_split);
  return want > 1 ? (int)want : 0;
}

// Launch plan: tile variant -> wide-row tile? -> KV splits (rules above) -> build and bias placement -> scratch.
Plan make_plan(const ffpa_fwd_params* p, const DimEntry* de) {
  Plan pl = {};
  // <= 32 query rows per (batch, head): one 32-row block per workgroup with D split over all four waves;
  // the reference switches to its split-KV decode kernels only for Nq == 1 (native/launch.cuh:306-340)
  pl.variant = (p->seqlen_q <= 32 && !(p->flags & FFPA_FLAG_DEBUG_SAFE_PATH)) ? 1 : 0;
  de->config(pl.variant, &pl.br, &pl.bc, &pl.lds);
  const int64_t cus = device_cu_count();
  pl.wide = pick_wide_tile(p, de, pl, cus) ? 1 : 0;
  if (pl.wide) de->config(3, &pl.br, &pl.bc, &pl.lds);
  pl.nqt = (p->seqlen_q + pl.br - 1) / pl.br;
  pl.nt = (p->seqlen_kv + pl.bc - 1) / pl.bc;
  const PlanCtx c = plan_context(p, de, &pl, cus);
  const int64_t want = (p->flags & FFPA_FLAG_TILE_RANGES) ? 0 : pick_splits(c);  // (the flag: this launch's ranges are per row tile or none — pick_tile_ranges below)
  pl.splits = want > 0 ? (int)clamp_splits(c, want) : 1;
  pl.m16 = (pl.variant == 0 && !(p->flags & FFPA_FLAG_DEBUG_SAFE_PATH) && c.dk >= FFPA_M16_MIN_D) ? 1 : 0;
  if (pl.m16) pick_m16_build(p, de, pl, c.dk);
  pick_small_d_bias_tiles(p, de, pl);
  pl.chunk = pick_dense_head_chunk(p, pl, c.dk);
  if (const int ranges = pick_tile_ranges(p, pl, c)) {
    pl.tile_ranges = 1;
    pl.splits = ranges;
    if (pl.chunk < 1) pl.chunk = 1;
  }
  pl.pair = (pl.chunk <= 1 && pick_pair_tiles(p, pl)) ? 1 : 0;
  pl.tiles_per_split = (pl.nt + pl.splits - 1) / pl.splits;
  if (pl.tiles_per_split < 1) pl.tiles_per_split = 1;
  if (!pl.tile_ranges) pl.splits = (pl.nt + pl.tiles_per_split - 1) / pl.tiles_per_split;  // (per-row-tile ranges: the kernel sizes them, every count is exact)
  if (pl.splits < 1) pl.splits = 1;
  pl.ws_bytes = pl.splits > 1 ? (size_t)pl.splits * p->batch * p->heads_q * p->seqlen_q * ((size_t)c.dk + 1) * sizeof(float) : 0;
  return pl;
}

// Shape / dtype checks shared by the launch and the queries.  Returns FFPA_OK or a status.
int check_basic(const ffpa_fwd_params* p, const DimEntry** de_out) {
  if (p == nullptr) return fail(FFPA_ERR_NULL_POINTER, "params is NULL");
  if (p->struct_size != sizeof(ffpa_fwd_params) || p->abi_version != FFPA_ATTN_ABI_VERSION)
    return fail(FFPA_ERR_BAD_ABI, "ffpa_fwd_params ABI mismatch: size %u (want %zu), version %u (want %d)",
                p->struct_size, sizeof(ffpa_fwd_params), p->abi_version, FFPA_ATTN_ABI_VERSION);
  if (p->batch <= 0 || p->heads_q <= 0 || p->heads_kv <= 0 || p->seqlen_q <= 0 || p->seqlen_kv <= 0)
    return fail(FFPA_ERR_BAD_SHAPE, "non-positive dimension: B=%d Hq=%d Hkv=%d Nq=%d Nkv=%d", p->batch, p->heads_q,
                p->heads_kv, p->seqlen_q, p->seqlen_kv);
  const DimEntry* de = find_dim(p->head_dim);
  if (de == nullptr)
    return fail(FFPA_ERR_BAD_HEADDIM, "headdim not support! D=%d (supported: multiples of 8 in [8, 1024])", p->head_dim);
  *de_out = de;
  return FFPA_OK;
}

// Dropout keeps the element of Philox word w when u(w) = ((float)w + 1.0f) * 2^-32 > p (prefill.cuh:437-440).  u is monotone in w
// (int -> float conversion and the add both round monotonically), so the decision is w >= T for the smallest kept word T — found here by
// bisection with the very float expression, once per call; the kernels compare integers.  p < 1 (checked) keeps w = 2^32 - 1 (u = 1).
uint32_t dropout_keep_threshold(float p) {
  if (!(p > 0.f)) return 0u;
  auto kept = [p](uint32_t w) { return ((float)w + 1.0f) * 2.3283064365386963e-10f > p; };
  uint32_t lo = 0u, hi = 0xFFFFFFFFu;  // kept(hi) holds; the answer is in [lo, hi]
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2u;
    if (kept(mid)) hi = mid; else lo = mid + 1u;
  }
  return lo;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int check_strides(const char* name, const int64_t s[3]) {
  for (int i = 0; i < 3; ++i) {
    if (s[i] < 0) return fail(FFPA_ERR_BAD_STRIDE, "%s stride[%d]=%lld is negative", name, i, (long long)s[i]);
    if (s[i] % 8 != 0)
      return fail(FFPA_ERR_BAD_STRIDE, "%s stride[%d]=%lld is not a multiple of 8 elements (16 bytes)", name, i,
                  (long long)s[i]);
  }
  return FFPA_OK;
}

}  // namespace

extern "C" {

int ffpa_attn_fwd(const ffpa_fwd_params* p, void* stream) {
  if (p == nullptr) return fail(FFPA_ERR_NULL_POINTER, "params is NULL");
  if (p->struct_size != sizeof(ffpa_fwd_params) || p->abi_version != FFPA_ATTN_ABI_VERSION)
    return fail(FFPA_ERR_BAD_ABI, "ffpa_fwd_params ABI mismatch: size %u (want %zu), version %u (want %d)",
                p->struct_size, sizeof(ffpa_fwd_params), p->abi_version, FFPA_ATTN_ABI_VERSION);
  if (!p->q || !p->k || !p->v || !p->o) return fail(FFPA_ERR_NULL_POINTER, "q/k/v/o must be non-NULL");
  if (p->dtype != FFPA_DTYPE_BF16 && p->dtype != FFPA_DTYPE_FP16)
    return fail(FFPA_ERR_BAD_DTYPE, "dtype %d is not bf16(0)/fp16(1)", p->dtype);
  if (p->batch <= 0 || p->heads_q <= 0 || p->heads_kv <= 0 || p->seqlen_q <= 0 || p->seqlen_kv <= 0)
    return fail(FFPA_ERR_BAD_SHAPE, "non-positive dimension: B=%d Hq=%d Hkv=%d Nq=%d Nkv=%d", p->batch, p->heads_q,
                p->heads_kv, p->seqlen_q, p->seqlen_kv);
  if (p->heads_q % p->heads_kv != 0)
    return fail(FFPA_ERR_BAD_SHAPE, "num_heads: Hq=%d is not a multiple of Hkv=%d", p->heads_q, p->heads_kv);
  const DimEntry* de = find_dim(p->head_dim);
  if (de == nullptr)
    return fail(FFPA_ERR_BAD_HEADDIM, "headdim not support! D=%d (supported: multiples of 8 in [8, 1024])", p->head_dim);
  if (!aligned16(p->q) || !aligned16(p->k) || !aligned16(p->v) || !aligned16(p->o))
    return fail(FFPA_ERR_MISALIGNED, "q/k/v/o base pointers must be 16-byte aligned");
  int rc;
  if ((rc = check_strides("q", p->q_stride)) || (rc = check_strides("k", p->k_stride)) ||
      (rc = check_strides("v", p->v_stride)) || (rc = check_strides("o", p->o_stride)))
    return rc;
  // K / V tiles are fetched with 32-bit buffer offsets relative to the tile's first row: one tile (<= 128
  // rows) must span < 4 GiB; the slice itself may be of any size
  for (const int64_t* st : {p->k_stride, p->v_stride}) {
    if (st[2] < p->head_dim || st[2] >= (1LL << 24))
      return fail(FFPA_ERR_BAD_STRIDE, "k/v row stride %lld: rows must not overlap and must be < 2^24 elements apart",
                  (long long)st[2]);
  }
  if ((p->bias == nullptr) != (p->bias_dtype == FFPA_BIAS_NONE))
    return fail(FFPA_ERR_BAD_DTYPE, "bias pointer and bias_dtype disagree (ptr %s, dtype %d)",
                p->bias ? "set" : "NULL", p->bias_dtype);
  if (p->bias_dtype < FFPA_BIAS_NONE || p->bias_dtype > FFPA_BIAS_BOOL8)
    return fail(FFPA_ERR_BAD_DTYPE, "unknown bias_dtype %d", p->bias_dtype);
  for (int i = 0; i < 4; ++i)
    if (p->bias && p->bias_stride[i] < 0) return fail(FFPA_ERR_BAD_STRIDE, "bias stride[%d] is negative", i);
  if (!(p->dropout_p >= 0.f && p->dropout_p < 1.f))
    return fail(FFPA_ERR_BAD_SHAPE, "dropout_p=%g must be in [0, 1)", (double)p->dropout_p);
  if (p->dropout_p > 0.f && p->causal_row_mod != 0)
    return fail(FFPA_ERR_UNSUPPORTED, "dropout with packed query heads (causal_row_mod) is not supported");
  if (!isfinite(p->softmax_scale)) return fail(FFPA_ERR_BAD_SHAPE, "softmax_scale is not finite");
  const int safe = (p->flags & FFPA_FLAG_DEBUG_SAFE_PATH) ? 1 : 0;

  if (p->causal_row_mod < 0) return fail(FFPA_ERR_BAD_SHAPE, "causal_row_mod must be >= 0");
  if (p->kv_bounds != nullptr && p->causal_row_mod != 0)
    return fail(FFPA_ERR_UNSUPPORTED, "kv_bounds with packed query heads (causal_row_mod) is not supported");
  if (p->kv_bounds != nullptr && (p->kv_bounds_stride[0] < 0 || p->kv_bounds_stride[1] < 0))
    return fail(FFPA_ERR_BAD_STRIDE, "kv_bounds strides must be >= 0");
  if (p->workspace != nullptr && !aligned16(p->workspace))
    return fail(FFPA_ERR_MISALIGNED, "workspace must be 16-byte aligned");
  if (p->split_tickets != nullptr && (reinterpret_cast<uintptr_t>(p->split_tickets) & 3u) != 0)
    return fail(FFPA_ERR_MISALIGNED, "split_tickets must be 4-byte aligned");
  if ((rc = check_device()) != FFPA_OK) return rc;
  const Plan pl = make_plan(p, de);
  const int lds = pl.lds;
  const int64_t nqt = pl.nqt;
  const int64_t grid = (int64_t)p->batch * p->heads_q * (pl.pair ? (nqt + 1) / 2 : nqt) * pl.splits;
  if (grid > 0x7fffffffLL) return fail(FFPA_ERR_BAD_SHAPE, "grid of %lld workgroups is too large", (long long)grid);

  ffpa::FwdArgs a;
  memset(&a, 0, sizeof(a));
  a.q = p->q;
  a.k = p->k;
  a.v = p->v;
  a.o = p->o;
  a.lse = p->lse;
  a.bias = p->bias;
  for (int i = 0; i < 3; ++i) {
    a.sq[i] = p->q_stride[i];
    a.sk[i] = p->k_stride[i];
    a.sv[i] = p->v_stride[i];
    a.so[i] = p->o_stride[i];
  }
  for (int i = 0; i < 4; ++i) a.sbias[i] = p->bias ? p->bias_stride[i] : 0;
  a.B = p->batch;
  a.Hq = p->heads_q;
  a.Hkv = p->heads_kv;
  a.Nq = p->seqlen_q;
  a.Nkv = p->seqlen_kv;
  a.d_valid = p->head_dim;
  a.group = p->heads_q / p->heads_kv;
  a.nqt = (int)nqt;
  a.pair_tiles = pl.pair;
  a.total_wg = (int)grid;
  a.bias_dtype = p->bias_dtype;
  a.causal = p->causal ? 1 : 0;
  a.causal_offset = p->causal_offset;
  a.scale_log2 = p->softmax_scale * 1.4426950408889634f;  // FFPA_M_LOG2E, csrc/cuffpa/common.cuh:9-18
  a.inv_scale = 1.f;
  if (pl.m16) {
    // the 16x16x32 build folds the scale into its exponent (needs scale > 0) and takes additive biases in units of 1 / scale: a zero scale
    // (scores = bias) reaches it as "Q = 0, scale = 1", a negative one as "-Q, |scale|" — the same scores
    if (p->softmax_scale == 0.f) {
      a.scale_log2 = 1.4426950408889634f;
      a.q_mode = 1;
    } else {
      const float mag = fabsf(p->softmax_scale);  // a negative scale: (-Q, |scale|) — exact, and the kernel's fused exponent needs scale > 0
      a.scale_log2 = mag * 1.4426950408889634f;
      a.inv_scale = (float)(1.0 / (double)mag);
      if (p->softmax_scale < 0.f) a.q_mode = 2;
    }
  }
  a.thr = p->rescale_threshold < 0.f ? 8.0f : p->rescale_threshold;
  a.flags = p->flags & 0x7fffffffu;  // (bit 31 is the launch side's own: kFlagStreamKV)
  {
    // Short-query launches stream K / V with the non-temporal hint when nothing would hit in a cache anyway: every (batch, kv head) is read by ONE
    // workgroup per KV range (MHA, or query heads packed into the rows by the caller: heads_q == heads_kv) and K + V are past the 256 MiB Infinity
    // Cache (a cache of that size read every decode step IS served from it: B4 H32 Nkv 1024 D512 = 256 MiB measured - 5 % with the hint, 128 MiB - 3 %;
    // 288 MiB + 3 %, 320 MiB + 5 ... 7 %, 384 / 512 MiB + 5 ... 8 %: the line is drawn at 272 MiB).  LDS-DMA from HBM: 5.9
    // TB/s without the hint, 7.3 with it (tools/probes/hbm_read_probe.hip); decode B1 H32 Nkv 8192 D512 102.8 -> 95.6 us per step, B8 GQA 180 -> 169,
    // D = 1024 179 -> 170, 131072 keys 356 -> 335; un-packed GQA (several workgroups read the same K / V through L2) LOSES 2 ... 15 % and keeps the
    // plain form (profiles/r04_kv_stream.txt).
    const int64_t kv_bytes = 2LL * p->batch * p->heads_kv * p->seqlen_kv * p->head_dim * 2;
    bool stream = pl.variant == 1 && p->heads_q == p->heads_kv && kv_bytes >= (272LL << 20);
    if (p->flags & FFPA_FLAG_KV_STREAM) stream = pl.variant == 1;
    if (p->flags & FFPA_FLAG_NO_KV_STREAM) stream = false;
    if (stream) a.flags |= ffpa::kFlagStreamKV;
  }
  a.nsplit = pl.splits;
  a.tiles_per_split = pl.tiles_per_split;
  a.causal_row_mod = p->causal_row_mod;
  a.kv_bounds = p->kv_bounds;
  a.s_bounds[0] = p->kv_bounds ? p->kv_bounds_stride[0] : 0;
  a.s_bounds[1] = p->kv_bounds ? p->kv_bounds_stride[1] : 0;
  if (p->bias != nullptr && p->bias_stride[3] == 1) {
    // vector bias loads (16 consecutive keys per lane): W elements per load need W-element aligned base and
    // batch / head / row strides.  16-byte loads when possible (W = 8 for 16-bit, 4 for fp32), else 8-byte.
    const int esz = p->bias_dtype == FFPA_BIAS_FP32 ? 4 : p->bias_dtype == FFPA_BIAS_BOOL8 ? 1 : 2;
    for (int w : {16 / esz, esz == 1 ? 16 : 4}) {  // boolean masks: 16-byte loads or byte loads
      bool ok = (reinterpret_cast<uintptr_t>(p->bias) % (size_t)(w * esz)) == 0;
      for (int i = 0; i < 3; ++i) ok = ok && (p->bias_stride[i] % w == 0);
      if (ok) {
        a.bias_vec = w;
        break;
      }
    }
  }
  // A key bias (no row axis, unit key stride, 16-byte aligned rows, 16- or 32-bit) is cached in LDS once per workgroup when the
  // head dim's tiles leave room for a whole number of tiles' worth of it (ffpa_common.h: FwdArgs.bias_lds)
  if (!pl.m16 && p->bias != nullptr && pl.variant == 0 && pl.splits == 1 && p->seqlen_q > 1 && p->bias_stride[2] == 0 && p->bias_stride[3] == 1 &&
      p->bias_dtype >= FFPA_BIAS_FP16 && p->bias_dtype <= FFPA_BIAS_FP32 && !(p->flags & FFPA_FLAG_NO_BIAS_LDS) && !safe) {
    const int esz = p->bias_dtype == FFPA_BIAS_FP32 ? 4 : 2;
    const int64_t row_bytes = (int64_t)p->seqlen_kv * esz;
    const int64_t bytes = (int64_t)pl.nt * pl.bc * esz;
    const bool ok = reinterpret_cast<uintptr_t>(p->bias) % 16 == 0 && row_bytes % 16 == 0 && (p->bias_stride[0] * esz) % 16 == 0 &&
                    (p->bias_stride[1] * esz) % 16 == 0 && pl.lds + bytes <= 160 * 1024;
    if (ok) a.bias_lds = (int)bytes;
  }
  if (pl.m16) {
    a.bias_tile = pl.btile;
    a.bias_lds = pl.bias_lds;
    a.bias_cache_raw = pl.bias_raw;
  } else if (pl.btile) {
    a.bias_tile = 1;
    a.bias_lds = -(4 * 32 * pl.bc * 2);  // negative: LDS bytes reserved for the tile staging (no key-bias row cache)
  }
  a.dropout_p = p->dropout_p;
  a.keep_scale = p->dropout_p > 0.f ? 1.f / (1.f - p->dropout_p) : 1.f;
  a.keep_threshold = dropout_keep_threshold(p->dropout_p);
  // XCDs per head (xcd_logical_id, ffpa_common.h).  One XCD per head keeps a head's K/V stream in one L2; but then eight heads are in flight chip-wide,
  // and once their K + V no longer fit the Infinity Cache the second and later rounds of a head's row tiles come from HBM instead (config 3: eight heads x
  // 32 MiB = the whole 256 MiB).  Prefill launches with at least two rounds of row tiles per head and XCD therefore share a head between the smallest
  // power-of-two number of XCDs that brings the K + V in flight under 200 MiB (measured, profiles/r03_xcd_group.txt).
  {
    int g = 1;
    const unsigned forced = (p->flags >> 8) & 7u;
    if (forced != 0) {
      g = 1 << (forced - 1 > 3 ? 3 : forced - 1);
    } else if (pl.m16 && pl.splits == 1 && (int64_t)pl.nqt * (p->heads_q / p->heads_kv) >= 64) {
      const double kv_mib = 2.0 * (double)p->seqlen_kv * (double)p->head_dim * 2.0 / 1048576.0;  // K + V of one (batch, kv head)
      while (g < 8 && (8 / g) * kv_mib > 200.0) g *= 2;
    }
    a.xcd_group = g;
  }
  // The split-D tiles (D > 512) give a DMA piece less than a microsecond to land (32-key steps, single K / V buffers): their launches touch the
  // tile two steps ahead (ffpa_fwd_m16_kernel.h, "L2 prefetch").  Measured, same library with and without (profiles/r03_l2_prefetch.txt):
  // D = 576 ... 1024: + 3 ... 9 % (D = 768: +- 0), every shape tried (self, cross, GQA, batch 4, causal, key bias); D <= 512: - 1 ... 2 %, off.
  a.l2_prefetch = (p->flags & FFPA_FLAG_NO_L2_PREFETCH) ? 0 : ((p->flags & FFPA_FLAG_L2_PREFETCH) || (pl.m16 && kernel_head_dim(p->head_dim) > 512)) ? 1 : 0;
  a.philox_seed = p->philox_seed;
  a.philox_offset = p->philox_offset;
  if (pl.splits > 1) {
    a.ws_o = static_cast<float*>(p->workspace);
    a.ws_lse = a.ws_o + (size_t)pl.splits * p->batch * p->heads_q * p->seqlen_q * kernel_head_dim(p->head_dim);
    // short-query launches with tickets: the kernel merges the partials itself (last split of a row tile to arrive).  Prefill tiles always
    // take the merge kernel: their row tiles are 64 - 128 rows, one workgroup merging a whole tile serialises what the merge kernel
    // spreads over thousands of workgroups (measured 2x slower: profiles/r03_split_merge.txt)
    a.tickets = pl.variant == 1 ? p->split_tickets : nullptr;
  }

  int st;
  if (pl.chunk > 1 || pl.tile_ranges) {
    // the packed-sequence kernel in its dense mode (no boundary arrays): this launch's FwdArgs as they are, that kernel's workgroup order — and, with KV ranges,
    // its per-row-tile ranges into the dense call's workspace rows [split, batch, head, row]
    ffpa::VarlenArgs va;
    memset(&va, 0, sizeof(va));
    va.lse_stride_h = p->seqlen_q;
    va.head_chunk = pl.chunk > 1 ? pl.chunk : 1;
    va.ws_head_rows = p->seqlen_q;
    va.ws_split_rows = (int64_t)p->batch * p->heads_q * p->seqlen_q;
    st = -3;
    for (const VarlenEntry& e : kVarlenDims)
      if (e.d == kernel_head_dim(p->head_dim)) st = e.launch(p->dtype, 0, a, va, static_cast<hipStream_t>(stream));
  } else {
    st = de->launch(p->dtype, safe, pl.wide ? 3 : pl.variant, a, static_cast<hipStream_t>(stream));
  }
  if (st == 0 && pl.splits > 1 && a.tickets == nullptr) {
    const unsigned rows = (unsigned)((int64_t)p->batch * p->heads_q * p->seqlen_q);
    if (p->dtype == FFPA_DTYPE_BF16)
      hipLaunchKernelGGL(ffpa::ffpa_fwd_merge_kernel<__bf16>, dim3(rows, (unsigned)(p->head_dim + 255) / 256), dim3(64), 0, static_cast<hipStream_t>(stream), a, kernel_head_dim(p->head_dim));
    else
      hipLaunchKernelGGL(ffpa::ffpa_fwd_merge_kernel<_Float16>, dim3(rows, (unsigned)(p->head_dim + 255) / 256), dim3(64), 0, static_cast<hipStream_t>(stream), a, kernel_head_dim(p->head_dim));
    st = (int)hipGetLastError();
  }
  if (st == -3) return fail(FFPA_ERR_UNSUPPORTED, "debug safe-path kernel is not built for D=%d / this dtype", p->head_dim);
  if (st == -2)
    return fail(FFPA_ERR_LAUNCH, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d) failed (is this a gfx950?)", lds);
  if (st < 0) return fail(FFPA_ERR_LAUNCH, "launch setup failed (%d)", st);
  if (st != 0)
    return fail(FFPA_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(st)));
  return FFPA_OK;
}

size_t ffpa_attn_fwd_workspace_bytes(const ffpa_fwd_params* params) {
  const DimEntry* de = nullptr;
  if (check_basic(params, &de) != FFPA_OK) return 0;
  // size for the split count the heuristic would pick with unlimited scratch
  ffpa_fwd_params q = *params;
  q.workspace = reinterpret_cast<void*>(16);
  q.workspace_bytes = ~0ull;
  return make_plan(&q, de).ws_bytes;
}

size_t ffpa_attn_fwd_split_tickets(const ffpa_fwd_params* params) {
  const DimEntry* de = nullptr;
  if (check_basic(params, &de) != FFPA_OK) return 0;
  ffpa_fwd_params q = *params;
  q.workspace = reinterpret_cast<void*>(16);
  q.workspace_bytes = ~0ull;
  const Plan pl = make_plan(&q, de);
  return (pl.splits > 1 && pl.variant == 1) ? (size_t)params->batch * params->heads_q * pl.nqt : 0;
}

int ffpa_attn_mask_kv_bounds(const void* bias, int bias_dtype, const int64_t bias_stride[4], int bb, int hb, int nq,
                             int nkv, int32_t* out, void* stream) {
  if (bias == nullptr || out == nullptr || bias_stride == nullptr) return fail(FFPA_ERR_NULL_POINTER, "bias / out / strides are NULL");
  if (bias_dtype < FFPA_BIAS_FP16 || bias_dtype > FFPA_BIAS_BOOL8) return fail(FFPA_ERR_BAD_DTYPE, "unknown bias_dtype %d", bias_dtype);
  if (bb <= 0 || hb <= 0 || nq <= 0 || nkv <= 0) return fail(FFPA_ERR_BAD_SHAPE, "non-positive mask dimension");
  for (int i = 0; i < 4; ++i)
    if (bias_stride[i] < 0) return fail(FFPA_ERR_BAD_STRIDE, "bias stride[%d] is negative", i);
  ffpa::MaskBoundsArgs m;
  m.bias = bias;
  for (int i = 0; i < 4; ++i) m.sb[i] = bias_stride[i];
  m.hb = hb;
  m.nq = nq;
  m.nkv = nkv;
  m.nblk = (nq + 31) / 32;
  m.words = (nkv + 31) / 32;
  if ((size_t)m.words * 4 > 60 * 1024) m.words = 0;  // the neutral-key bitmap must fit LDS (Nkv <= 491520); else no free range
  const size_t smem = (size_t)m.words * 4;
  m.out = out;
  const int64_t grid = (int64_t)bb * hb * m.nblk;
  if (grid > 0x7fffffffLL) return fail(FFPA_ERR_BAD_SHAPE, "mask of %lld row blocks is too large", (long long)grid);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int w = bias_dtype == FFPA_BIAS_FP32 ? 4 : bias_dtype == FFPA_BIAS_BOOL8 ? 16 : 8;  // elements per 16-byte load
  const bool vec = bias_stride[3] == 1 && nkv % w == 0 && reinterpret_cast<uintptr_t>(bias) % 16 == 0 &&
                   bias_stride[0] % w == 0 && bias_stride[1] % w == 0 && bias_stride[2] % w == 0;
  const dim3 g((unsigned)grid), blk(256);
  if (vec) {
    if (bias_dtype == FFPA_BIAS_FP32) hipLaunchKernelGGL(ffpa::ffpa_mask_kv_bounds_vec_kernel<float>, g, blk, smem, st, m);
    else if (bias_dtype == FFPA_BIAS_BOOL8) hipLaunchKernelGGL(ffpa::ffpa_mask_kv_bounds_vec_kernel<uint8_t>, g, blk, smem, st, m);
    else if (bias_dtype == FFPA_BIAS_BF16) hipLaunchKernelGGL(ffpa::ffpa_mask_kv_bounds_vec_kernel<__bf16>, g, blk, smem, st, m);
    else hipLaunchKernelGGL(ffpa::ffpa_mask_kv_bounds_vec_kernel<_Float16>, g, blk, smem, st, m);
  } else {
    if (bias_dtype == FFPA_BIAS_FP32) hipLaunchKernelGGL(ffpa::ffpa_mask_kv_bounds_kernel<float>, g, blk, smem, st, m);
    else if (bias_dtype == FFPA_BIAS_BOOL8) hipLaunchKernelGGL(ffpa::ffpa_mask_kv_bounds_kernel<uint8_t>, g, blk, smem, st, m);
    else if (bias_dtype == FFPA_BIAS_BF16) hipLaunchKernelGGL(ffpa::ffpa_mask_kv_bounds_kernel<__bf16>, g, blk, smem, st, m);
    else hipLaunchKernelGGL(ffpa::ffpa_mask_kv_bounds_kernel<_Float16>, g, blk, smem, st, m);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(FFPA_ERR_LAUNCH, "mask bounds launch failed: %s", hipGetErrorString(e));
  return FFPA_OK;
}

int ffpa_attn_fwd_plan(const ffpa_fwd_params* params, int out[4]) {
  const DimEntry* de = nullptr;
  const int rc = check_basic(params, &de);
  if (rc != FFPA_OK) return rc;
  if (out == nullptr) return fail(FFPA_ERR_NULL_POINTER, "out is NULL");
  const Plan pl = make_plan(params, de);
  out[0] = pl.variant;
  out[1] = pl.br;
  out[2] = pl.bc;
  out[3] = pl.splits;
  return FFPA_OK;
}

int ffpa_attn_fwd_kernel(const ffpa_fwd_params* params, char* buf, size_t n) {
  const DimEntry* de = nullptr;
  const int rc = check_basic(params, &de);
  if (rc != FFPA_OK) return rc;
  if (buf == nullptr || n == 0) return fail(FFPA_ERR_NULL_POINTER, "buf is NULL / empty");
  const Plan pl = make_plan(params, de);
  const char* dt = params->dtype == FFPA_DTYPE_FP16 ? "fp16" : "bf16";
  const int drop = params->dropout_p > 0.f ? 1 : 0;
  const char* merge = pl.splits > 1 ? ((params->split_tickets != nullptr && pl.variant == 1) ? " (in-launch split merge)" : " + ffpa_fwd_merge_kernel") : "";
  if (pl.wide) {
    snprintf(buf, n, "ffpa_fwd_m16w_kernel<%s, %d, RH=%d, MK=%d>%s", dt, de->d, pl.br / 64, pl.mk, merge);
  } else if (pl.tile_ranges) {
    snprintf(buf, n, "ffpa_fwd_m16_varlen_kernel<%s, %d> (dense launch, head chunks of %d, KV ranges per row tile)%s", dt, de->d, pl.chunk > 1 ? pl.chunk : 1, merge);
  } else if (pl.chunk > 1) {
    snprintf(buf, n, "ffpa_fwd_m16_varlen_kernel<%s, %d> (dense launch, head chunks of %d)", dt, de->d, pl.chunk);
  } else if (pl.m16) {
    snprintf(buf, n, "ffpa_fwd_m16_kernel<%s, %d, MK=%d, DROP=%d%s>%s", dt, de->d, pl.mk, drop, pl.pair ? ", PAIR" : "", merge);
  } else {
    const int nd = pl.variant == 1 ? ((de->d % 128 == 0) ? 4 : 2) : (de->d <= 512 ? 1 : 2);
    snprintf(buf, n, "ffpa_fwd_split_d_kernel<%s, %d, ND=%d%s%s%s>%s", dt, de->d, nd, (params->flags & FFPA_FLAG_DEBUG_SAFE_PATH) ? ", SAFE" : "",
             drop ? ", DROP" : "", pl.btile ? ", BTILE" : "", merge);
  }
  return FFPA_OK;
}

// ---- packed sequences (include/ffpa_attn.h: ffpa_varlen_fwd_params)
namespace {

struct VarlenPlan {
  const VarlenEntry* ve;
  int br, bc, nqt;
  int pack;  // > 0: decode batch under GQA — the query heads of a KV group are the rows of the tile (VarlenArgs::pack)
  int nt;    // 1: the build whose K / V pieces carry the non-temporal hint
  int splits;        // KV ranges per sequence (1 = the KV axis is not split)
  size_t ws_bytes;   // scratch the split launch uses
  int64_t grid;      // workgroups of the launch (all ranges)
  int compact;       // > 0: row-tile slots per head of the compact grid (VarlenArgs::compact_tiles)
};

// Validation shared by the launch and the queries; fills the plan.  Head dims below the first 16x16x32 instantiation run on it (columns past the
// caller's head dim read as zeros and are never stored, as in the dense call).
int varlen_plan(const ffpa_varlen_fwd_params* p, VarlenPlan* out) {
  if (p == nullptr) return fail(FFPA_ERR_NULL_POINTER, "params is NULL");
  if (p->struct_size != sizeof(ffpa_varlen_fwd_params) || p->abi_version != FFPA_ATTN_ABI_VERSION)
    return fail(FFPA_ERR_BAD_ABI, "ffpa_varlen_fwd_params ABI mismatch: size %u (want %zu), version %u (want %d)", p->struct_size,
                sizeof(ffpa_varlen_fwd_params), p->abi_version, FFPA_ATTN_ABI_VERSION);
  if (p->dtype != FFPA_DTYPE_BF16 && p->dtype != FFPA_DTYPE_FP16) return fail(FFPA_ERR_BAD_DTYPE, "dtype %d is not bf16(0)/fp16(1)", p->dtype);
  if (p->batch <= 0 || p->heads_q <= 0 || p->heads_kv <= 0 || p->max_seqlen_q <= 0 || p->max_seqlen_kv < 0)
    return fail(FFPA_ERR_BAD_SHAPE, "non-positive dimension: batch=%d Hq=%d Hkv=%d max_seqlen_q=%d max_seqlen_kv=%d", p->batch, p->heads_q, p->heads_kv,
                p->max_seqlen_q, p->max_seqlen_kv);
  if (p->heads_q % p->heads_kv != 0) return fail(FFPA_ERR_BAD_SHAPE, "num_heads: Hq=%d is not a multiple of Hkv=%d", p->heads_q, p->heads_kv);
  if (p->head_dim <= 0 || p->head_dim % 8 != 0 || p->head_dim > 1024)
    return fail(FFPA_ERR_BAD_HEADDIM, "headdim not support! D=%d (supported: multiples of 8 in [8, 1024])", p->head_dim);
  int dk = kernel_head_dim(p->head_dim);
  if (dk < FFPA_M16_MIN_D) dk = FFPA_M16_MIN_D;
  const VarlenEntry* ve = nullptr;
  for (const VarlenEntry& e : kVarlenDims)
    if (e.d == dk) ve = &e;
  if (ve == nullptr) return fail(FFPA_ERR_BAD_HEADDIM, "headdim not support! D=%d", p->head_dim);
  out->ve = ve;
  out->br = 128 / (dk <= 512 ? 1 : 2);
  out->bc = ffpa::m16_block_keys(dk, false);
  // short query sequences under GQA — decode (one token per sequence), speculative decoding / multi-token prediction / small prefill chunks (a few) —: the group's
  // heads x the sequence's tokens ride in the rows of ONE tile per (sequence, KV head), head-major (VarlenArgs::pack) — the K / V stream of a group is read once, not
  // once per query head, and the tile's MFMA rows are `group` times better used.  Packed when every sequence's rows fit one tile: group x max_seqlen_q <= block rows.
  // Measured (tools/gpu_varlen_decode.py, profiles/r06_varlen_pack_tokens.txt): 16 sequences x 16 tokens, Hq 32 / Hkv 8, D 512: 2.2 -> 4.7 TB/s of K + V; 2.1 ... 5.2 x one workgroup per query head on five shapes, bit-identical.
  const int group = p->heads_q / p->heads_kv;
  out->pack = (group > 1 && (int64_t)group * p->max_seqlen_q <= out->br && !(p->flags & FFPA_FLAG_NO_PACK_GQA)) ? group : 0;
  out->nqt = out->pack ? 1 : (p->max_seqlen_q + out->br - 1) / out->br;
  out->grid = (int64_t)p->batch * (out->pack ? p->heads_kv : p->heads_q) * out->nqt;
  // The COMPACT grid: a caller that says how many rows q has (total_q, ABI 6; >= cu_seqlens_q[batch]) lets a ragged prefill batch size its grid by the rows there are —
  // sum_i ceil(len_i / block rows) <= ceil(total_q / block rows) + batch slots per head — instead of batch x the longest sequence's row tiles, most of which would find
  // no row and leave (each of them still holds a CU for a launch and a few loads).  The kernel finds a slot's (sequence, row tile) on the device — a scan every
  // workgroup pays —, so it is taken when at least three quarters of the full grid would be idle: one 16k-token prompt among 63 short ones 7454 -> 7039 us (+ 6 %),
  // 8192 + 31 short + 4 %, D = 128 one 8k among 127 short ones + 46 %; the bench batch (8 sequences of 256 ... 4864: 55 % idle) - 1 % and not taken
  // (tools/gpu_varlen_compact.py, profiles/r06_varlen_compact_grid.txt).  FFPA_FLAG_NO_COMPACT_GRID keeps the full grid (same order, same bits).
  out->compact = 0;
  if (!out->pack && out->nqt > 1 && p->total_q > 0 && !(p->flags & FFPA_FLAG_NO_COMPACT_GRID)) {
    const int64_t slots = ((int64_t)p->total_q + out->br - 1) / out->br + p->batch;
    if (slots * 4 <= (int64_t)p->batch * out->nqt && slots <= 0x7fffffffLL) {
      out->compact = (int)slots;
      out->grid = slots * p->heads_q;
    }
  }
  // The non-temporal K / V fetch (the dense short-query launches' rule, ffpa_attn_fwd): every K / V byte is read by ONE workgroup — one row tile per (sequence,
  // head), and MHA or packed GQA rows — and the batch's K + V do not fit the 256 MiB Infinity Cache.  The launch side sees only max_seqlen_kv, not the lengths: it
  // prices a ragged batch at half of batch x max (>= 272 MiB of that).  LDS-DMA from HBM: 5.9 TB/s without the hint, 7.3 with it (profiles/r04_kv_stream.txt);
  // packed decode batches: profiles/r06_varlen.txt.  FFPA_FLAG_KV_STREAM / _NO_KV_STREAM force either.
  {
    const bool one_reader = out->nqt == 1 && (out->pack > 0 || group == 1);
    const int64_t kv_bound = 2LL * p->batch * p->heads_kv * (int64_t)p->max_seqlen_kv * p->head_dim * 2;
    bool nt = one_reader && kv_bound / 2 >= (272LL << 20);
    if (p->flags & FFPA_FLAG_KV_STREAM) nt = true;
    if (p->flags & FFPA_FLAG_NO_KV_STREAM) nt = false;
    out->nt = nt ? 1 : 0;
  }
  // KV SPLITS (ABI 6).  A launch of one row tile per (sequence, head) — decode, speculative decoding, chunked prefill — splits every sequence's KV range over `splits`
  // workgroups (the sequence's own tiles / splits per range, computed on the device) + ffpa_varlen_merge_kernel.  The launch side sees max_seqlen_kv only.  Two
  // reasons, measured on packed decode batches (tools/gpu_varlen_splits.py, profiles/r06_varlen_splits.txt):
  //   (a) FILL the chip — the dense short-query launches' rule (pick_splits): one workgroup per CU for head dims >= 320 (a tile pair is 128 KiB of LDS there), else
  //       two; at least kT.min_tiles_short KV tiles of the longest sequence per range.  8 sequences x 8 KV heads = 64 workgroups: 649 -> 212 us with 4 ranges;
  //   (b) BALANCE a ragged batch — several sequences differ in length and the longest one's workgroups finish last: up to kT.varlen_balance_wgs workgroups per slot
  //       in ranges of at least kT.varlen_balance_min_tiles tiles (256 pairs x 4 ranges: 877 -> 728 us; 64 x 8: 181 us), as long as the partials stay a small
  //       fraction of the K + V bytes (64 query rows per sequence: 4 ranges 248 us, 16 ranges 293 us).
  //   (c) PREFILL launches (several row tiles per head) that leave most of the chip idle — chunked prefill of one long sequence with a few heads per GPU: 8 heads x
  //       8 row tiles are 64 workgroups — take the dense path's under-filled rule (pick_splits): fill the slots, at least kT.min_tiles_prefill KV tiles per
  //       range; under the causal flag every row tile shares out the tiles up to ITS diagonal (the kernel), and the launch side prices the average row tile
  //       (max_seqlen_kv - max_seqlen_q / 2 keys).  64 workgroups x 4 ranges: 679 -> 242 us (392 -> 1100 TFLOPS); 16 x 16: 2600 -> 231 us.
  out->splits = 1;
  out->ws_bytes = 0;
  if (p->total_q > 0 && p->workspace != nullptr && p->num_splits != 1 && !(p->flags & FFPA_FLAG_DETERMINISTIC)) {
    const int64_t cus = device_cu_count();
    const int64_t max_tiles = ((int64_t)p->max_seqlen_kv + out->bc - 1) / out->bc;
    int64_t want = 1;
    if ((p->flags & FFPA_FLAG_FORCE_SPLITS) && p->num_splits > 1) {
      want = p->num_splits < max_tiles ? p->num_splits : max_tiles;  // (sweeps and tests: exactly n, down to one tile per range)
    } else if (out->nqt > 1) {
      // (c): slots as in (a); the workgroups that find rows — the grid is sized by max_seqlen_q, a ragged batch's short sequences leave theirs at once — are at
      // least heads x ceil(total_q / block rows) (exact when the lengths are whole tiles; a range too many costs little, one too few a lot: 2 sequences of 1024 /
      // 256 rows, 8 heads: 2 ranges 361 us, 3 270, 4 277).  A causal launch that fills the slots only once and whose longest row tile walks >= 1.5 x the average
      // one (a whole prompt: 1 ... n tiles) takes two ranges: the second half of the long rows fills the slots the short ones leave (8 heads x 4096 x 4096:
      // 187 -> 168 us; D = 128: 78 -> 70).  profiles/r06_varlen_prefill_splits.txt
      const int64_t slots = (dk >= kT.short_one_per_cu_min_d ? 1 : 2) * cus;
      int64_t live = (int64_t)p->heads_q * (((int64_t)p->total_q + out->br - 1) / out->br);
      if (live > out->grid) live = out->grid;
      if (live < 1) live = 1;
      if (2 * live <= slots) want = slots / live;
      double avg_tiles = (double)max_tiles;
      if (p->causal) {
        const double mq = p->max_seqlen_q < p->max_seqlen_kv ? p->max_seqlen_q : p->max_seqlen_kv;  // (rows past the keys see none)
        avg_tiles = ((double)p->max_seqlen_kv - 0.5 * mq) / out->bc;
        if (want == 1 && live <= slots && (double)max_tiles >= 1.5 * avg_tiles) want = 2;
      }
      const int64_t cap = (int64_t)(avg_tiles / kT.min_tiles_prefill);
      if (want > cap) want = cap;
      if (want < 1) want = 1;
      if (p->num_splits > 1 && want > p->num_splits) want = p->num_splits;
      const double per_split = (double)p->heads_q * (double)p->total_q * ((double)dk + 1.0) * sizeof(float);
      if ((double)want * per_split > kT.max_auto_ws_bytes) want = (int64_t)(kT.max_auto_ws_bytes / per_split);
    } else {
      const int64_t slots = (dk >= kT.short_one_per_cu_min_d ? 1 : 2) * cus;
      int64_t fill = 2 * out->grid <= slots ? slots / out->grid : 1;
      if (fill > max_tiles / kT.min_tiles_short) fill = max_tiles / kT.min_tiles_short;
      int64_t balance = 1;
      if (p->batch > 1 && out->grid < kT.varlen_balance_wgs * slots) {
        balance = kT.varlen_balance_wgs * slots / out->grid;
        if (balance > max_tiles / kT.varlen_balance_min_tiles) balance = max_tiles / kT.varlen_balance_min_tiles;
        // partials: splits x Hq x total_q x D x 4 B, written and read back; K + V: every pair streams its sequence — priced at 0.6 x the longest (a ragged batch)
        const double kv_bytes = (double)out->grid * 0.6 * (double)p->max_seqlen_kv * dk * 4.0;
        const double per_split_rw = 2.0 * (double)p->heads_q * (double)p->total_q * dk * 4.0;
        const int64_t afford = (int64_t)(kT.varlen_partial_frac * kv_bytes / per_split_rw);
        if (balance > afford) balance = afford;
      }
      want = fill > balance ? fill : balance;
      if (want < 1) want = 1;
      if (p->num_splits > 1 && want > p->num_splits) want = p->num_splits;
    }
    if (want > ffpa::kMergeMaxSplits) want = ffpa::kMergeMaxSplits;
    const size_t per_split = (size_t)p->heads_q * (size_t)p->total_q * ((size_t)dk + 1) * sizeof(float);
    if ((uint64_t)want * per_split > p->workspace_bytes) want = (int64_t)(p->workspace_bytes / per_split);
    if (want > 1 && out->grid * want <= 0x7fffffffLL) {
      out->splits = (int)want;
      out->ws_bytes = (size_t)want * per_split;
      out->grid *= want;
    }
  }
  if (out->grid > 0x7fffffffLL) return fail(FFPA_ERR_BAD_SHAPE, "grid of %lld workgroups is too large", (long long)out->grid);
  return FFPA_OK;
}

int check_strides2(const char* name, const int64_t s[2]) {
  for (int i = 0; i < 2; ++i) {
    if (s[i] < 0) return fail(FFPA_ERR_BAD_STRIDE, "%s stride[%d]=%lld is negative", name, i, (long long)s[i]);
    if (s[i] % 8 != 0) return fail(FFPA_ERR_BAD_STRIDE, "%s stride[%d]=%lld is not a multiple of 8 elements (16 bytes)", name, i, (long long)s[i]);
  }
  return FFPA_OK;
}

}  // namespace

int ffpa_attn_varlen_fwd(const ffpa_varlen_fwd_params* p, void* stream) {
  VarlenPlan pl;
  int rc = varlen_plan(p, &pl);
  if (rc != FFPA_OK) return rc;
  if (!p->q || !p->k || !p->v || !p->o) return fail(FFPA_ERR_NULL_POINTER, "q/k/v/o must be non-NULL");
  if (!p->cu_seqlens_q || !p->cu_seqlens_kv) return fail(FFPA_ERR_NULL_POINTER, "cu_seqlens_q / cu_seqlens_kv must be non-NULL");
  if ((reinterpret_cast<uintptr_t>(p->cu_seqlens_q) & 3u) || (reinterpret_cast<uintptr_t>(p->cu_seqlens_kv) & 3u) || (reinterpret_cast<uintptr_t>(p->seqused_kv) & 3u))
    return fail(FFPA_ERR_MISALIGNED, "cu_seqlens_q / cu_seqlens_kv / seqused_kv must be 4-byte aligned");
  if (!aligned16(p->q) || !aligned16(p->k) || !aligned16(p->v) || !aligned16(p->o))
    return fail(FFPA_ERR_MISALIGNED, "q/k/v/o base pointers must be 16-byte aligned");
  if ((rc = check_strides2("q", p->q_stride)) || (rc = check_strides2("k", p->k_stride)) || (rc = check_strides2("v", p->v_stride)) ||
      (rc = check_strides2("o", p->o_stride)))
    return rc;
  for (const int64_t* st : {p->k_stride, p->v_stride}) {
    if (st[0] < p->head_dim || st[0] >= (1LL << 24))
      return fail(FFPA_ERR_BAD_STRIDE, "k/v row stride %lld: rows must not overlap and must be < 2^24 elements apart", (long long)st[0]);
  }
  if (p->lse != nullptr && p->lse_stride_head < 0) return fail(FFPA_ERR_BAD_STRIDE, "lse_stride_head is negative");
  if (!isfinite(p->softmax_scale)) return fail(FFPA_ERR_BAD_SHAPE, "softmax_scale is not finite");
  if (p->total_q < 0) return fail(FFPA_ERR_BAD_SHAPE, "total_q=%d is negative", p->total_q);
  if (p->workspace != nullptr && !aligned16(p->workspace)) return fail(FFPA_ERR_MISALIGNED, "workspace must be 16-byte aligned");
  if ((rc = check_device()) != FFPA_OK) return rc;

  ffpa::FwdArgs a;
  memset(&a, 0, sizeof(a));
  a.q = p->q;
  a.k = p->k;
  a.v = p->v;
  a.o = p->o;
  a.lse = p->lse;
  // element strides batch / head / row: a sequence's base is its row offset (the kernel adds it), so the batch stride is zero
  a.sq[1] = p->q_stride[1], a.sq[2] = p->q_stride[0];
  a.sk[1] = p->k_stride[1], a.sk[2] = p->k_stride[0];
  a.sv[1] = p->v_stride[1], a.sv[2] = p->v_stride[0];
  a.so[1] = p->o_stride[1], a.so[2] = p->o_stride[0];
  a.B = p->batch;
  a.Hq = p->heads_q;
  a.Hkv = p->heads_kv;
  a.Nq = p->max_seqlen_q;   // (the kernel replaces both by the sequence's own)
  a.Nkv = p->max_seqlen_kv;
  a.d_valid = p->head_dim;
  a.group = p->heads_q / p->heads_kv;
  a.nqt = pl.nqt;
  a.total_wg = (int)pl.grid;
  a.causal = p->causal ? 1 : 0;
  // the 16x16x32 build folds the scale into its exponent (needs scale > 0): a zero scale reaches it as "Q = 0, scale = 1", a negative one as "-Q, |scale|" (ffpa_attn_fwd)
  a.inv_scale = 1.f;
  if (p->softmax_scale == 0.f) {
    a.scale_log2 = 1.4426950408889634f;
    a.q_mode = 1;
  } else {
    const float mag = fabsf(p->softmax_scale);
    a.scale_log2 = mag * 1.4426950408889634f;
    a.inv_scale = (float)(1.0 / (double)mag);
    if (p->softmax_scale < 0.f) a.q_mode = 2;
  }
  a.thr = p->rescale_threshold < 0.f ? 8.0f : p->rescale_threshold;
  a.flags = p->flags & (FFPA_FLAG_NO_XCD_REMAP);
  a.nsplit = 1;
  a.tiles_per_split = 0x7fffffff / 2;  // (never the binding limit: the KV axis is not split)
  a.keep_scale = 1.f;
  {
    // XCDs per head: the dense call's rule (ffpa_attn_fwd), priced on the longest sequence the caller announces
    int g = 1;
    const unsigned forced = (p->flags >> 8) & 7u;
    if (forced != 0) {
      g = 1 << (forced - 1 > 3 ? 3 : forced - 1);
    } else if ((int64_t)pl.nqt * (p->heads_q / p->heads_kv) >= 64) {
      const double kv_mib = 2.0 * (double)p->max_seqlen_kv * (double)p->head_dim * 2.0 / 1048576.0;
      while (g < 8 && (8 / g) * kv_mib > 200.0) g *= 2;
    }
    a.xcd_group = g;
  }
  a.l2_prefetch = (p->flags & FFPA_FLAG_NO_L2_PREFETCH) ? 0 : ((p->flags & FFPA_FLAG_L2_PREFETCH) || pl.ve->d > 512) ? 1 : 0;

  ffpa::VarlenArgs va;
  va.cu_q = p->cu_seqlens_q;
  va.cu_k = p->cu_seqlens_kv;
  va.lse_stride_h = p->lse_stride_head;
  // heads of one KV group that walk a sequence side by side (ffpa_fwd_m16_varlen_kernel; pick_head_chunk: whole chunks per XCD, never across KV groups — MHA and
  // head counts that do not divide into eight chunks run plain head-major order)
  va.head_chunk = pick_head_chunk(p->heads_q, p->heads_q / p->heads_kv);
  va.pack = pl.pack;
  va.used_k = p->seqused_kv;
  va.q_tok_stride = p->q_stride[0];
  va.o_tok_stride = p->o_stride[0];
  if (pl.pack) {
    // FwdArgs describes Hkv heads of `pack` rows: head stride = one KV group, row stride = one query head
    a.Hq = p->heads_kv;
    a.group = 1;
    a.sq[1] = p->q_stride[1] * pl.pack, a.sq[2] = p->q_stride[1];
    a.so[1] = p->o_stride[1] * pl.pack, a.so[2] = p->o_stride[1];
    if (p->max_seqlen_q == 1) a.causal = 0;  // (a single token sees every key of its sequence; more tokens: the kernel sets causal_row_mod per sequence)
    va.head_chunk = 1;  // (the rows of a tile ARE the group: KV heads share nothing)
  }

  va.compact_tiles = pl.compact;
  va.ws_head_rows = va.ws_split_rows = 0;
  if (pl.splits > 1) {
    // partials [split, query head, token, Dk] fp32 + their LSE [split, query head, token]; the kernel finds a range's tiles from its sequence's own length
    a.nsplit = pl.splits;
    a.ws_o = static_cast<float*>(p->workspace);
    a.ws_lse = a.ws_o + (size_t)pl.splits * p->heads_q * p->total_q * pl.ve->d;
    va.ws_head_rows = p->total_q;
    va.ws_split_rows = (int64_t)p->heads_q * p->total_q;
  }

  int st = pl.ve->launch(p->dtype, pl.nt, a, va, static_cast<hipStream_t>(stream));
  if (st == 0 && pl.splits > 1) {
    const dim3 grid((unsigned)((int64_t)p->heads_q * p->total_q), (unsigned)(p->head_dim + 255) / 256);
    if (p->dtype == FFPA_DTYPE_BF16)
      hipLaunchKernelGGL(ffpa::ffpa_varlen_merge_kernel<__bf16>, grid, dim3(64), 0, static_cast<hipStream_t>(stream), a, va, pl.ve->d, p->batch, p->o_stride[1]);
    else
      hipLaunchKernelGGL(ffpa::ffpa_varlen_merge_kernel<_Float16>, grid, dim3(64), 0, static_cast<hipStream_t>(stream), a, va, pl.ve->d, p->batch, p->o_stride[1]);
    st = (int)hipGetLastError();
  }
  if (st == -2) return fail(FFPA_ERR_LAUNCH, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed (is this a gfx950?)");
  if (st < 0) return fail(FFPA_ERR_LAUNCH, "launch setup failed (%d)", st);
  if (st != 0) return fail(FFPA_ERR_LAUNCH, "kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(st)));
  return FFPA_OK;
}

int ffpa_attn_varlen_fwd_plan(const ffpa_varlen_fwd_params* params, int out[5]) {
  VarlenPlan pl;
  const int rc = varlen_plan(params, &pl);
  if (rc != FFPA_OK) return rc;
  if (out == nullptr) return fail(FFPA_ERR_NULL_POINTER, "out is NULL");
  out[0] = pl.nqt;
  out[1] = pl.br;
  out[2] = pl.bc;
  out[3] = (int)pl.grid;
  out[4] = pl.splits;
  return FFPA_OK;
}

size_t ffpa_attn_varlen_fwd_workspace_bytes(const ffpa_varlen_fwd_params* params) {
  if (params == nullptr || params->struct_size != sizeof(ffpa_varlen_fwd_params)) return 0;
  // size for the split count the heuristic would pick with unlimited scratch
  ffpa_varlen_fwd_params q = *params;
  q.workspace = reinterpret_cast<void*>(16);
  q.workspace_bytes = ~0ull;
  VarlenPlan pl;
  if (varlen_plan(&q, &pl) != FFPA_OK) return 0;
  return pl.ws_bytes;
}

int ffpa_attn_varlen_fwd_kernel(const ffpa_varlen_fwd_params* params, char* buf, size_t n) {
  VarlenPlan pl;
  const int rc = varlen_plan(params, &pl);
  if (rc != FFPA_OK) return rc;
  if (buf == nullptr || n == 0) return fail(FFPA_ERR_NULL_POINTER, "buf is NULL");
  snprintf(buf, n, "ffpa_fwd_m16_varlen_kernel<%s, %d%s>%s%s", params->dtype == FFPA_DTYPE_BF16 ? "bf16" : "fp16", pl.ve->d, pl.nt ? ", NT" : "",
           pl.pack ? " (GQA heads packed into rows)" : "", pl.splits > 1 ? " + ffpa_varlen_merge_kernel" : "");
  return FFPA_OK;
}

int ffpa_attn_query(int what) {
  switch (what) {
    case FFPA_QUERY_ABI_VERSION: return FFPA_ATTN_ABI_VERSION;
    case FFPA_QUERY_FWD_AVAILABLE: return 1;
    case FFPA_QUERY_MIN_HEAD_DIM: return 8;
    case FFPA_QUERY_MAX_HEAD_DIM: return 1024;
    case FFPA_QUERY_HEAD_DIM_MULTIPLE: return 8;
    case FFPA_QUERY_FP16_AVAILABLE: return 1;
    case FFPA_QUERY_DROPOUT_AVAILABLE: return 1;
#ifdef FFPA_INST_SAFE
    case FFPA_QUERY_DEBUG_KERNELS: return 1;  // the test-only twin library
#else
    case FFPA_QUERY_DEBUG_KERNELS: return 0;
#endif
    // what the launch plan's pricing reads from the current device (fallbacks without one): compute units, engine clock, HBM peak
    case FFPA_QUERY_DEVICE_CUS: return device_cu_count();
    case FFPA_QUERY_DEVICE_CLOCK_MHZ: return (int)(device_rates().cu_flops / 4096.0 / 1e6);
    case FFPA_QUERY_DEVICE_HBM_GBPS: return (int)(device_rates().hbm_bytes / 1e9);
    case FFPA_QUERY_VARLEN_AVAILABLE: return 1;
    default: return -1;
  }
}

int ffpa_attn_fwd_tile_config(int head_dim, int* block_rows, int* block_keys, int* lds_bytes) {
  const DimEntry* de = find_dim(head_dim);
  if (de == nullptr) return fail(FFPA_ERR_BAD_HEADDIM, "headdim not support! D=%d", head_dim);
  int br = 0, bc = 0, lds = 0;
  de->config(0, &br, &bc, &lds);
  if (block_rows) *block_rows = br;
  if (block_keys) *block_keys = bc;
  if (lds_bytes) *lds_bytes = lds;
  return FFPA_OK;
}

const char* ffpa_attn_last_error(void) { return g_err; }

#ifdef FFPA_PRODUCT_BUILD
const char* ffpa_attn_version(void) { return "ffpa-attn-amd 0.5.0 gfx950"; }
#else
const char* ffpa_attn_version(void) { return "ffpa-attn-amd 0.5.0 gfx950 (developer variant: NOT a product build)"; }
#endif

}  // extern "C"
