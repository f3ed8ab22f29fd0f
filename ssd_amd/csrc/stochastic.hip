// Temperature > 0: sampling and ratio-based speculative verification.
//  * ssd_sample_rows    Sampler.forward (reference ssd/layers/sampler.py:15-36): temperature 0 -> argmax, else a draw
//                       from softmax(logits / T).  The reference draws argmax(p / Exp(1)); that is the Gumbel-max trick,
//                       so here it is argmax(logit / T + Gumbel) in ONE pass, no softmax, no V-sized temporaries.
//  * ssd_row_lse        log-sum-exp of logits / T per row (the softmax normaliser verify() needs, verify.py:76-99).
//  * ssd_verify_ratio   verify() for rows with temperature > 0 (reference ssd/utils/verify.py:50-167):
//                       accept x_i with probability min(1, p_i(x_i) / q_i(x_i)); at the first rejection n draw the
//                       recovery token from normalise(max(0, p_n - q_n)); if everything was accepted (or the row is not
//                       a "ratio row": cache miss without JIT) draw it from p_n; temperature-0 rows take the greedy branch.
// Randomness is a counter-based hash of (seed word in device memory, stream salt, row, index): launches inside a
// hipGraph stay fresh because the seed word is advanced on the device (ssd_rng_advance) after every use.
#include "common.h"

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
// uniform in (0, 1), 24 bits
__device__ __forceinline__ float u01(uint64_t seed, uint32_t salt, uint32_t row, uint32_t idx) {
  const uint64_t h = mix64(seed ^ mix64(((uint64_t)salt << 40) ^ ((uint64_t)row << 20 << 12) ^ idx));
  return ((float)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float gumbel(float u) { return -__logf(-__logf(u)); }

struct Best { float v; int i; };
__device__ __forceinline__ Best bmax(Best a, Best b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }

template <int THREADS>
__device__ Best block_best(Best best, Best* sm) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = bmax(best, Best{__shfl_xor(best.v, o, 64), __shfl_xor(best.i, o, 64)});
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = best;
  __syncthreads();
  Best r = sm[0];
  for (int w = 1; w < THREADS / 64; ++w) r = bmax(r, sm[w]);
  return r;
}
template <int THREADS>
__device__ float block_sum(float v, float* sm) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  for (int w = 0; w < THREADS / 64; ++w) r += sm[w];
  return r;
}
template <int THREADS>
__device__ float block_max(float v, float* sm) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sm[0];
  for (int w = 1; w < THREADS / 64; ++w) r = fmaxf(r, sm[w]);
  return r;
}

constexpr int ST_THREADS = 1024;
constexpr int ST_MAXBOOST = 8;      // sampler_x rescales the F+1 largest probabilities; F+1 <= 8

// sampler_x (reference apply_sampler_x_rescaling, async_spec_helpers.py:79-105): the F+1 most probable tokens of a row
// get their probability multiplied by x before renormalisation.  In log space that is "+ log x" on those logits: a Gumbel
// draw and a log-sum-exp over the boosted logits are exactly the rescaled, renormalised distribution.
struct Boost {
  int idx[ST_MAXBOOST];
  int k;
  float log_x;
  __device__ __forceinline__ float of(int i) const {
    float b = 0.f;
#pragma unroll
    for (int j = 0; j < ST_MAXBOOST; ++j) b = (j < k && idx[j] == i) ? log_x : b;
    return b;
  }
};
__device__ __forceinline__ Boost load_boost(const int32_t* __restrict__ boost_idx, int row, int k, float log_x) {
  Boost b;
  b.k = boost_idx ? k : 0;
  b.log_x = log_x;
#pragma unroll
  for (int j = 0; j < ST_MAXBOOST; ++j) b.idx[j] = (boost_idx && j < k) ? boost_idx[(size_t)row * k + j] : -1;
  return b;
}

// out[row][0..k) = indices of the k largest logits of the row, largest first, lowest index on ties
__global__ void __launch_bounds__(ST_THREADS)
topk_rows_kernel(const bf16_t* __restrict__ logits, long ld, int V, int k, int32_t* __restrict__ out) {
  __shared__ Best sm[ST_THREADS / 64];
  __shared__ int chosen[ST_MAXBOOST];
  const int row = blockIdx.x;
  const bf16_t* x = logits + (size_t)row * ld;
  for (int r = 0; r < k; ++r) {
    Best best = {-INFINITY, 0x7fffffff};
    for (int i = threadIdx.x; i < V; i += ST_THREADS) {
      bool taken = false;
      for (int j = 0; j < r; ++j) taken = taken || chosen[j] == i;
      if (!taken) best = bmax(best, Best{bf2f(x[i]), i});
    }
    best = block_best<ST_THREADS>(best, sm);
    if (threadIdx.x == 0) { chosen[r] = best.i; out[(size_t)row * k + r] = best.i; }
    __syncthreads();
  }
}

extern "C" int ssd_topk_rows(const void* logits, long ld, int T, int V, int k, int32_t* out, void* stream) {
  if (T <= 0 || V <= 0 || k < 1 || k > ST_MAXBOOST || k > V) return SSD_ERR_SHAPE;
  hipLaunchKernelGGL(topk_rows_kernel, dim3(T), dim3(ST_THREADS), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, V, k, out);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

__global__ void __launch_bounds__(ST_THREADS)
sample_rows_kernel(const bf16_t* __restrict__ logits, long ld, int V, const float* __restrict__ temps, int rows_per_temp,
                   const uint64_t* __restrict__ rng, uint32_t salt, int64_t* __restrict__ out, int64_t* __restrict__ out2,
                   const int32_t* __restrict__ boost_idx, int boost_k, float log_x) {
  __shared__ Best sm[ST_THREADS / 64];
  const int row = blockIdx.x;
  const float T = temps[row / rows_per_temp];
  const Boost bo = load_boost(boost_idx, row, boost_k, log_x);
  const bf16_t* x = logits + (size_t)row * ld;
  const uint64_t seed = *rng;
  Best best = {-INFINITY, 0x7fffffff};
  if (T == 0.f) {
    for (int i = threadIdx.x; i < V; i += ST_THREADS) best = bmax(best, Best{bf2f(x[i]), i});
  } else {
    const float inv = 1.0f / fmaxf(T, 1e-8f);
    for (int i = threadIdx.x; i < V; i += ST_THREADS)
      best = bmax(best, Best{bf2f(x[i]) * inv + bo.of(i) + gumbel(u01(seed, salt, row, i)), i});
  }
  best = block_best<ST_THREADS>(best, sm);
  if (threadIdx.x == 0) { out[row] = best.i; if (out2) out2[row] = best.i; }
}

extern "C" int ssd_sample_rows(const void* logits, long ld, int T, int V, const float* temps, int rows_per_temp,
                               const void* rng_state, unsigned salt, int64_t* out, int64_t* out2, const int32_t* boost_idx,
                               int boost_k, float boost_x, void* stream) {
  if (T <= 0 || V <= 0 || rows_per_temp <= 0) return SSD_ERR_SHAPE;
  if (boost_idx && (boost_k < 1 || boost_k > ST_MAXBOOST || !(boost_x > 0.f))) return SSD_ERR_ARG;
  hipLaunchKernelGGL(sample_rows_kernel, dim3(T), dim3(ST_THREADS), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, V,
                     temps, rows_per_temp, (const uint64_t*)rng_state, salt, out, out2, boost_idx, boost_k,
                     boost_idx ? logf(boost_x) : 0.f);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

__global__ void rng_advance_kernel(uint64_t* rng) { *rng = mix64(*rng + 0x632be59bd9b4e019ull); }
extern "C" int ssd_rng_advance(void* rng_state, void* stream) {
  if (!rng_state) return SSD_ERR_ARG;
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (uint64_t*)rng_state);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

// lse[row] = log(sum_v exp(logit_v / T - m)) + m with m = max_v logit_v / T  (rows with T == 0 get lse = +inf marker 0)
__global__ void __launch_bounds__(ST_THREADS)
row_lse_kernel(const bf16_t* __restrict__ logits, long ld, int V, const float* __restrict__ temps, int rows_per_temp,
               float* __restrict__ lse, const int32_t* __restrict__ boost_idx, int boost_k, float log_x) {
  __shared__ float sm[ST_THREADS / 64];
  const int row = blockIdx.x;
  const float T = temps[row / rows_per_temp];
  if (T == 0.f) { if (threadIdx.x == 0) lse[row] = 0.f; return; }
  const float inv = 1.0f / fmaxf(T, 1e-8f);
  const bf16_t* x = logits + (size_t)row * ld;
  const Boost bo = load_boost(boost_idx, row, boost_k, log_x);
  float m = -INFINITY;
  for (int i = threadIdx.x; i < V; i += ST_THREADS) m = fmaxf(m, bf2f(x[i]) * inv + bo.of(i));
  m = block_max<ST_THREADS>(m, sm);
  float s = 0.f;
  for (int i = threadIdx.x; i < V; i += ST_THREADS) s += __expf(bf2f(x[i]) * inv + bo.of(i) - m);
  s = block_sum<ST_THREADS>(s, sm);
  if (threadIdx.x == 0) lse[row] = __logf(s) + m;
}

extern "C" int ssd_row_lse(const void* logits, long ld, int T, int V, const float* temps, int rows_per_temp, float* lse,
                           const int32_t* boost_idx, int boost_k, float boost_x, void* stream) {
  if (T <= 0 || V <= 0 || rows_per_temp <= 0) return SSD_ERR_SHAPE;
  if (boost_idx && (boost_k < 1 || boost_k > ST_MAXBOOST || !(boost_x > 0.f))) return SSD_ERR_ARG;
  hipLaunchKernelGGL(row_lse_kernel, dim3(T), dim3(ST_THREADS), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, V, temps,
                     rows_per_temp, lse, boost_idx, boost_k, boost_idx ? logf(boost_x) : 0.f);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

// One workgroup per sequence.
//   logits_p [B][K+1][ld_p], logits_q [B][K][ld_q]; spec [B][K+1] = (recovery, x_1..x_K); preds_p [B][K+1] = argmax rows of p;
//   lse_p [B][K+1], lse_q [B][K] from ssd_row_lse; temps_t / temps_q [B]; ratio_rows int32[B] (1: the draft tokens really
//   were sampled from q -- cache hit, or JIT speculation; 0: fall back to greedy acceptance and draw the recovery from p).
// Outputs as ssd_verify_greedy (+ accept_prob [B][K] for inspection when not null).
__global__ void __launch_bounds__(ST_THREADS)
verify_ratio_kernel(const bf16_t* __restrict__ lp, long ld_p, const bf16_t* __restrict__ lq, long ld_q, int V, int K,
                    const int64_t* __restrict__ spec, const int64_t* __restrict__ preds_p, const float* __restrict__ lse_p,
                    const float* __restrict__ lse_q, const float* __restrict__ temps_t, const float* __restrict__ temps_q,
                    const int32_t* __restrict__ ratio_rows, const uint64_t* __restrict__ rng, uint32_t salt,
                    int32_t* __restrict__ accept_len, int64_t* __restrict__ recovery, int64_t* __restrict__ packed,
                    float* __restrict__ accept_prob, const int32_t* __restrict__ boost_idx_q, int boost_k, float log_x) {
  __shared__ Best smb[ST_THREADS / 64];
  __shared__ float smf[ST_THREADS / 64];
  __shared__ int s_n;
  const int b = blockIdx.x;
  const float Tt = temps_t[b], Tq = temps_q[b];
  const uint64_t seed = *rng;
  const int64_t* sp = spec + (size_t)b * (K + 1);
  const int64_t* pr = preds_p + (size_t)b * (K + 1);
  const bool ratio = (Tt > 0.f || Tq > 0.f) && ratio_rows[b] != 0;
  const float inv_t = 1.0f / fmaxf(Tt, 1e-8f), inv_q = 1.0f / fmaxf(Tq, 1e-8f);
  // ---- acceptance: one lane per draft position, first rejection by ballot ----
  if (threadIdx.x < 64) {
    const int i = threadIdx.x;
    bool reject = false;
    if (i < K) {
      const int x = (int)sp[i + 1];
      if (ratio) {
        // p_i(x): softmax(logits_p / Tt) (one-hot at the argmax when Tt == 0); q_i(x) likewise
        const float pv = Tt > 0.f ? __expf(bf2f(lp[((size_t)b * (K + 1) + i) * ld_p + x]) * inv_t - lse_p[b * (K + 1) + i])
                                  : (pr[i] == x ? 1.f : 0.f);
        float qv;
        if (Tq > 0.f) {   // lse_q already carries the sampler_x boost (ssd_row_lse with the same boost rows)
          const Boost bo = load_boost(boost_idx_q, b * K + i, boost_k, log_x);
          qv = __expf(bf2f(lq[((size_t)b * K + i) * ld_q + x]) * inv_q + bo.of(x) - lse_q[b * K + i]);
        }
        else qv = 1.f;   // a greedy draft proposed its own argmax: q is one-hot at x
        const float a = fminf(pv / (qv + 1e-10f), 1.0f);
        if (accept_prob) accept_prob[b * K + i] = a;
        reject = !(u01(seed, salt, b, 1000 + i) <= a);
      } else {
        if (accept_prob) accept_prob[b * K + i] = (pr[i] == x) ? 1.f : 0.f;
        reject = pr[i] != x;
      }
    }
    const unsigned long long mask = __ballot(reject);
    if (i == 0) s_n = mask ? (int)__builtin_ctzll(mask) : K;
  }
  __syncthreads();
  const int n = s_n;
  // ---- recovery token ----
  int64_t rec;
  if (Tt == 0.f) {
    rec = pr[n];                                    // greedy target: argmax of row n
  } else {
    const bf16_t* prow = lp + ((size_t)b * (K + 1) + n) * ld_p;
    const float lsep = lse_p[b * (K + 1) + n];
    const bool adjust = ratio && n < K;
    Best best = {-INFINITY, 0x7fffffff};
    bool use_p = !adjust;
    if (adjust) {
      // residual r_v = max(0, p_v - q_v); if it vanishes everywhere fall back to p (verify.py:153-155)
      const bf16_t* qrow = lq + ((size_t)b * K + n) * ld_q;
      const float lseq = lse_q[b * K + n];
      const int xq = (int)sp[n + 1];
      const Boost bo = load_boost(boost_idx_q, b * K + n, boost_k, log_x);
      float tot = 0.f;
      for (int v = threadIdx.x; v < V; v += ST_THREADS) {
        const float pv = __expf(bf2f(prow[v]) * inv_t - lsep);
        const float qv = Tq > 0.f ? __expf(bf2f(qrow[v]) * inv_q + bo.of(v) - lseq) : (v == xq ? 1.f : 0.f);
        const float r = fmaxf(pv - qv, 0.f);
        tot += r;
        if (r > 0.f) best = bmax(best, Best{__logf(r) + gumbel(u01(seed, salt, b, 5000 + v)), v});
      }
      tot = block_sum<ST_THREADS>(tot, smf);
      if (!(tot > 0.f)) use_p = true;
    }
    if (use_p) {
      best = Best{-INFINITY, 0x7fffffff};
      for (int v = threadIdx.x; v < V; v += ST_THREADS)
        best = bmax(best, Best{bf2f(prow[v]) * inv_t + gumbel(u01(seed, salt, b, 5000 + v)), v});
    }
    best = block_best<ST_THREADS>(best, smb);
    rec = best.i;
  }
  if (threadIdx.x == 0) {
    accept_len[b] = n;
    recovery[b] = rec;
  }
  if (packed) {
    int64_t* row = packed + (size_t)b * (K + 3);
    if (threadIdx.x == 0) { row[0] = n; row[1] = rec; }
    if (threadIdx.x <= K) row[2 + threadIdx.x] = sp[threadIdx.x];
  }
}

extern "C" int ssd_verify_ratio(const void* logits_p, long ld_p, const void* logits_q, long ld_q, int V, int B, int K,
                                const int64_t* spec, const int64_t* preds_p, const float* lse_p, const float* lse_q,
                                const float* temps_t, const float* temps_q, const int32_t* ratio_rows, const void* rng_state,
                                unsigned salt, int32_t* accept_len, int64_t* recovery, int64_t* packed, float* accept_prob,
                                const int32_t* boost_idx_q, int boost_k, float boost_x, void* stream) {
  if (B <= 0 || K < 1 || K > 62 || V <= 0) return SSD_ERR_SHAPE;
  if (boost_idx_q && (boost_k < 1 || boost_k > ST_MAXBOOST || !(boost_x > 0.f))) return SSD_ERR_ARG;
  hipLaunchKernelGGL(verify_ratio_kernel, dim3(B), dim3(ST_THREADS), 0, (hipStream_t)stream, (const bf16_t*)logits_p, ld_p,
                     (const bf16_t*)logits_q, ld_q, V, K, spec, preds_p, lse_p, lse_q, temps_t, temps_q, ratio_rows,
                     (const uint64_t*)rng_state, salt, accept_len, recovery, packed, accept_prob, boost_idx_q, boost_k,
                     boost_idx_q ? logf(boost_x) : 0.f);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}
