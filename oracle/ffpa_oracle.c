/*
 * ffpa_oracle.c — CPU restatement of the reference's Split-D attention forward.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or executed by the
 * product path (ffpa_attn_amd/): only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it, and only as the checker.
 *
 * What it restates (the arithmetic, per query row; a row never interacts with another row):
 *   csrc/cuffpa/native/sm_80/split_d.cuh:222-228   causal KV-tile bounds (tail aligned, offset = Nkv - Nq)
 *   csrc/cuffpa/native/sm_80/split_d.cuh:506-539   per-tile order: KV-tail mask -> causal mask -> bias -> softmax
 *   csrc/cuffpa/native/prefill.cuh:548-555         bias enters as  scale*S + bias
 *   csrc/cuffpa/native/prefill.cuh:671-776         online softmax in the log2 domain, exp2, lazy rescale:
 *                                                  keep the stale max while it grew by <= FFPA_RESCALE_THRESHOLD
 *                                                  (= 8.0 log2 units, csrc/cuffpa/common.cuh:14)
 *   csrc/cuffpa/native/prefill.cuh:755-762         row sum from the UNROUNDED fp32 P; P rounded to bf16/fp16 (RN)
 *                                                  before the P.V contraction; fp32 accumulation
 *   csrc/cuffpa/native/prefill.cuh:877-1011        alpha = 2^(m - m'), l <- alpha*l + sum(P), O <- alpha*O + P16.V
 *   csrc/cuffpa/native/prefill.cuh:1018-1056       O_out = round(O * (1/l))
 *   csrc/cuffpa/native/prefill.cuh:1063-1073       LSE = ln(l) + m*ln2  (natural log)
 * Fully masked rows give NaN (exp2(-inf - -inf) in the reference; 0 * inf here) — same as SDPA.
 *
 * One deliberate simplification: the exponent's argument is formed as (s*c) - m in two fp32 roundings where the reference's kernel
 * (prefill.cuh:748) and the HIP kernel use one FMA.  The difference is at most one fp32 ulp of the score — three orders of
 * magnitude below the bf16 / fp16 rounding of P that every comparison against this oracle allows for.
 *
 * Parity pinning: the reference holds no golden vectors for this path (every forward test is
 * "allclose to PyTorch SDPA on seeded randn", tests/test_ffpa_fwd.py:106-121), so this oracle is
 * pinned (tests/test_oracle.py) against (i) OUTPUTS OF THE REFERENCE'S OWN large-head-dim kernel executed in the authoring
 * container — its Triton forward under TRITON_INTERPRET=1, fp16 and bf16, eleven cases incl. late score spikes that walk the
 * lazy-rescale branch and two 128-row x 8192-key cases at D = 512 (the headline key count) (tests/golden/make_triton_golden.py ->
 * ref_triton_cases.npz).  Qualification: the fp16 cases run the interpreter unmodified; the bf16 cases depend on two functions of
 * Triton's INTERPRETER being wrapped in the generator (its dot product multiplies bf16 bit patterns as integers and its fp32 -> bf16
 * cast truncates: the wrappers widen dot operands to fp32 and round to nearest even — make_triton_golden.py:63-83).  The reference's
 * kernel source is imported unchanged, but the bf16 pin is only as good as those twenty lines; (i') the same executed reference WITH
 * DROPOUT (round 4: its Triton forward carries the CUDA kernels' Philox mapping, triton/_ffpa_fwd.py:80-123) — three cases, fp16 and bf16,
 * offsets inside a Philox quad, batch / head terms, a 62-bit seed (make_triton_golden.py --dropout -> ref_triton_dropout.npz): the
 * stream, the element offsets, the keep rule and the 1 / (1 - p) scaling below are the executed reference's; (i'') the reference's
 * split-KV decode kernels (stage 1 per KV chunk + stage 2 LSE merge, triton/_ffpa_fwd.py:497-861) executed with 3 / 4 / 5 splits
 * (make_triton_golden.py --decode -> ref_triton_decode.npz): this oracle's single walk lands on the merged O and LSE; (i''') the reference's
 * FFPAAttnMeta.normalize in front of that kernel on user-level arguments (--api -> ref_triton_api.npz), (ii) PyTorch CPU SDPA — the reference's own
 * test oracle — on the committed fixtures in tests/golden/, (iii) the output of the reference's ffpa_attn_func
 * itself, run in the authoring container on config 1 (tests/golden/make_golden.py), and (iv) an
 * fp64 plain-math evaluation.
 *
 * Build:  gcc -O3 -march=native -fopenmp -shared -fPIC ffpa_oracle.c -o libffpa_oracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float bf16_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

static inline uint16_t f32_to_bf16(float f) { /* round to nearest even */
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40); /* NaN */
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

static inline float f16_to_f32(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t u;
  if (exp == 0) {
    if (man == 0) {
      u = sign;
    } else { /* subnormal */
      int e = -1;
      do {
        e++;
        man <<= 1;
      } while ((man & 0x400u) == 0);
      man &= 0x3ffu;
      u = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    u = sign | 0x7f800000u | (man << 13);
  } else {
    u = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &u, 4);
  return f;
}

static inline uint16_t f32_to_f16(float f) { /* round to nearest even, overflow -> inf */
  uint32_t u;
  memcpy(&u, &f, 4);
  uint32_t sign = (u >> 16) & 0x8000u;
  uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
  if (a >= 0x47800000u) return (uint16_t)(sign | 0x7c00u); /* >= 65536 -> inf (65520 rounds up below) */
  if (a < 0x33000001u) return (uint16_t)sign;               /* < 2^-25 -> 0 */
  int e = (int)(a >> 23) - 127;
  uint32_t man = (a & 0x7fffffu) | 0x800000u;
  int shift;
  uint32_t hexp;
  if (e < -14) { /* subnormal half */
    shift = 13 + (-14 - e);
    hexp = 0;
  } else {
    shift = 13;
    hexp = (uint32_t)(e + 15);
  }
  uint32_t halfway = 1u << (shift - 1);
  uint32_t rem = man & ((1u << shift) - 1u);
  uint32_t q = man >> shift;
  if (rem > halfway || (rem == halfway && (q & 1u))) q++;
  uint32_t h;
  if (hexp == 0) {
    h = q; /* may carry into the exponent: that is the right encoding */
  } else {
    h = ((hexp - 1) << 10) + q; /* q has the implicit bit: + 0x400 */
  }
  if (h >= 0x7c00u) h = 0x7c00u;
  return (uint16_t)(sign | h);
}

static inline float load_elem(const uint16_t* p, int dtype) { return dtype == 0 ? bf16_to_f32(*p) : f16_to_f32(*p); }
static inline uint16_t store_elem(float f, int dtype) { return dtype == 0 ? f32_to_bf16(f) : f32_to_f16(f); }
static inline float round_elem(float f, int dtype) { return load_elem(&(uint16_t){store_elem(f, dtype)}, dtype); }

/* Philox4x32-10 on counter (quad_lo, quad_hi, 0, 0), key = seed (csrc/cuffpa/native/prefill.cuh:398-422). */
void ffpa_oracle_philox(uint64_t seed, uint64_t quad, uint32_t out[4]) {
  uint32_t c0 = (uint32_t)quad, c1 = (uint32_t)(quad >> 32), c2 = 0u, c3 = 0u;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int round = 0; round < 10; ++round) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

/* keep iff u > p with u = (word + 1) * 2^-32 in fp32 (prefill.cuh:437-452,508-546) */
static inline int dropout_keeps(uint64_t seed, uint64_t element, float p) {
  uint32_t w[4];
  ffpa_oracle_philox(seed, element >> 2, w);
  const float u = ((float)w[element & 3] + 1.0f) * 2.3283064365386963e-10f;
  return u > p;
}

#define FFPA_LOG2E 1.4426950408889634f
#define FFPA_LN2 0.6931471805599453f

/*
 * Dense contiguous tensors: q [B,Hq,Nq,D], k/v [B,Hkv,Nkv,D], o [B,Hq,Nq,D] (16-bit, dtype 0 = bf16,
 * 1 = fp16), lse [B,Hq,Nq] fp32 (may be NULL), o_f32 [B,Hq,Nq,D] unrounded output (may be NULL).
 * bias: fp32 values or NULL, element strides bias_stride[4] = {batch, head, row, key} (0 = broadcast).
 * Rows row_begin <= r < row_end of every (batch, head) are computed; others are left untouched.
 * block_keys = KV tile length of the recurrence (the reference uses 128; the gfx950 kernel 64 / 32):
 * it only changes fp32 summation order and when the lazy rescale fires.
 * Returns 0, or -1 on bad arguments / allocation failure.
 */
/* pmax [B,Hq,2,Nq] (may be NULL): per row, plane 0 = the largest normalised probability max_k P / l, plane 1 = sum_k (P / l)^2.  Not part of
 * the reference's outputs — they are what a checker needs to bound (0) how far ONE differently-rounded P entry can move an output element
 * (2^-8 * pmax * |v| for bf16) and (1) the noise the 16-bit rounding of ALL P entries leaves in it (rms 2^-9 / sqrt 3 * sqrt(sum (P/l)^2) * rms |v|),
 * which two implementations that round P against different running maxima do not share. */
int ffpa_oracle_fwd_ex(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* o, float* o_f32, float* lse,
                       const float* bias, const int64_t* bias_stride, int B, int Hq, int Hkv, int Nq, int Nkv, int D,
                       int dtype, float scale, int causal, int causal_offset, float thr, int block_keys,
                       int row_begin, int row_end, float dropout_p, uint64_t philox_seed, uint64_t philox_offset, float* pmax) {
  if (!q || !k || !v || !o || B <= 0 || Hq <= 0 || Hkv <= 0 || Nq <= 0 || Nkv <= 0 || D <= 0) return -1;
  if (Hq % Hkv != 0 || block_keys <= 0 || (dtype != 0 && dtype != 1)) return -1;
  if (row_begin < 0) row_begin = 0;
  if (row_end > Nq) row_end = Nq;
  const int group = Hq / Hkv; /* split_d.cuh:135-136: kv head = q head / group */
  const float c = scale * FFPA_LOG2E;
  const float keep_scale = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
  int status = 0;

  for (int b = 0; b < B; ++b) {
    for (int hkv = 0; hkv < Hkv; ++hkv) {
      /* widen this kv head once */
      float* kf = (float*)malloc(sizeof(float) * (size_t)Nkv * D);
      float* vf = (float*)malloc(sizeof(float) * (size_t)Nkv * D);
      if (!kf || !vf) {
        free(kf);
        free(vf);
        return -1;
      }
      const uint16_t* kp = k + ((size_t)b * Hkv + hkv) * (size_t)Nkv * D;
      const uint16_t* vp = v + ((size_t)b * Hkv + hkv) * (size_t)Nkv * D;
      for (size_t i = 0; i < (size_t)Nkv * D; ++i) {
        kf[i] = load_elem(kp + i, dtype);
        vf[i] = load_elem(vp + i, dtype);
      }
      for (int g = 0; g < group; ++g) {
        const int hq = hkv * group + g;
#pragma omp parallel for schedule(dynamic, 4)
        for (int r = row_begin; r < row_end; ++r) {
          float* qf = (float*)malloc(sizeof(float) * D);
          float* acc = (float*)calloc((size_t)D, sizeof(float));
          float* x = (float*)malloc(sizeof(float) * block_keys);
          if (!qf || !acc || !x) {
            status = -1;
            free(qf);
            free(acc);
            free(x);
            continue;
          }
          const size_t qoff = (((size_t)b * Hq + hq) * Nq + r) * (size_t)D;
          for (int d = 0; d < D; ++d) qf[d] = load_elem(q + qoff + d, dtype);
          float m = -INFINITY, l = 0.f, xmax_all = -INFINITY;
          double l2 = 0.0; /* sum of P^2 (checker statistics only) */
          /* visible keys: key <= r + causal_offset (and key < Nkv) */
          long lim = causal ? (long)r + causal_offset : (long)Nkv - 1;
          if (lim > Nkv - 1) lim = Nkv - 1;
          int nt = (Nkv + block_keys - 1) / block_keys;
          for (int t = 0; t < nt; ++t) {
            const int k0 = t * block_keys;
            /* split_d.cuh:225-228 skips tiles past the causal diagonal of the CTA; per row that is
             * the same as processing an all-masked tile (P = 0, no state change). */
            if (causal && (long)k0 > lim) break;
            const int kn = (k0 + block_keys <= Nkv) ? block_keys : Nkv - k0;
            float tmax = -INFINITY;
            for (int j = 0; j < kn; ++j) {
              const int key = k0 + j;
              float s = 0.f;
              const float* kr = kf + (size_t)key * D;
              for (int d = 0; d < D; ++d) s += qf[d] * kr[d];
              float xv = s * c;
              if (bias) {
                const float bv = bias[(size_t)b * bias_stride[0] + (size_t)hq * bias_stride[1] +
                                      (size_t)r * bias_stride[2] + (size_t)key * bias_stride[3]];
                xv += bv * FFPA_LOG2E;
              }
              if ((long)key > lim) xv = -INFINITY;
              x[j] = xv;
              if (xv > tmax) tmax = xv;
            }
            if (tmax > xmax_all) xmax_all = tmax;
            const float m_new = (tmax > m) ? tmax : m;
            float alpha = 1.f;
            if (m_new > m + thr) { /* lazy rescale (prefill.cuh:684-755); first finite max always rescales */
              alpha = exp2f(m - m_new); /* m = -inf -> 0: O and l are still 0 */
              m = m_new;
            }
            const float m_use = (m == -INFINITY) ? 0.f : m;
            if (alpha != 1.f) {
              for (int d = 0; d < D; ++d) acc[d] *= alpha;
              l *= alpha;
              l2 *= (double)alpha * alpha;
            }
            float psum = 0.f;
            for (int j = 0; j < kn; ++j) {
              const float p = exp2f(x[j] - m_use);
              psum += p;
              l2 += (double)p * p;
              float p16 = round_elem(p, dtype);
              if (dropout_p > 0.f) { /* on the rounded P, after the row sum; rounded again (prefill.cuh:508-546) */
                const uint64_t e = philox_offset + (((uint64_t)b * Hq + hq) * Nq + (uint64_t)r) * (uint64_t)Nkv + (uint64_t)(k0 + j);
                p16 = dropout_keeps(philox_seed, e, dropout_p) ? round_elem(p16 * keep_scale, dtype) : 0.f;
              }
              if (p16 != 0.f) {
                const float* vr = vf + (size_t)(k0 + j) * D;
                for (int d = 0; d < D; ++d) acc[d] += p16 * vr[d];
              }
            }
            l += psum;
          }
          const float inv = 1.f / l;
          for (int d = 0; d < D; ++d) {
            const float val = acc[d] * inv;
            if (o_f32) o_f32[qoff + d] = val;
            o[qoff + d] = store_elem(val, dtype);
          }
          if (lse) lse[((size_t)b * Hq + hq) * Nq + r] = logf(l) + m * FFPA_LN2;
          if (pmax) {
            pmax[(((size_t)b * Hq + hq) * 2 + 0) * Nq + r] = exp2f(xmax_all - ((m == -INFINITY) ? 0.f : m)) * inv;
            pmax[(((size_t)b * Hq + hq) * 2 + 1) * Nq + r] = (float)(l2 * (double)inv * inv);
          }
          free(qf);
          free(acc);
          free(x);
        }
      }
      free(kf);
      free(vf);
    }
  }
  return status;
}

int ffpa_oracle_fwd_dropout(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* o, float* o_f32, float* lse,
                            const float* bias, const int64_t* bias_stride, int B, int Hq, int Hkv, int Nq, int Nkv, int D,
                            int dtype, float scale, int causal, int causal_offset, float thr, int block_keys,
                            int row_begin, int row_end, float dropout_p, uint64_t philox_seed, uint64_t philox_offset) {
  return ffpa_oracle_fwd_ex(q, k, v, o, o_f32, lse, bias, bias_stride, B, Hq, Hkv, Nq, Nkv, D, dtype, scale, causal, causal_offset, thr,
                            block_keys, row_begin, row_end, dropout_p, philox_seed, philox_offset, NULL);
}

int ffpa_oracle_fwd(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* o, float* o_f32, float* lse,
                    const float* bias, const int64_t* bias_stride, int B, int Hq, int Hkv, int Nq, int Nkv, int D,
                    int dtype, float scale, int causal, int causal_offset, float thr, int block_keys, int row_begin,
                    int row_end) {
  return ffpa_oracle_fwd_dropout(q, k, v, o, o_f32, lse, bias, bias_stride, B, Hq, Hkv, Nq, Nkv, D, dtype, scale, causal,
                                 causal_offset, thr, block_keys, row_begin, row_end, 0.f, 0, 0);
}

int ffpa_oracle_abi_version(void) { return 3; }
