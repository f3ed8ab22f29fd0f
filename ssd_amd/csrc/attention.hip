// Paged attention for every decode-side shape of the draft->verify loop, plus varlen causal prefill:
//   * single-query decode           (reference ssd/layers/attention.py:126-131, flash_attn_with_kvcache)
//   * K+1-query verify / glue       (reference ssd/layers/attention.py:105-111, bottom-right causal)
//   * branched draft-tree decode    (reference ssd/layers/attention.py:113-125 flashinfer custom mask;
//                                    mask definition ssd/engine/helpers/mask_helpers.py:12-21) -- here the
//                                    mask is STRUCTURAL (computed from branch index / step), no mask
//                                    tensor, no plan(), graph-capturable.
//   * varlen causal prefill         (reference ssd/layers/attention.py:90-93) -- reads the just-stored
//                                    K/V through the page table, which also covers prefix-cache hits.
// One wave per (16*RT query rows of one kv-head group, key range); a workgroup may hold up to 8 such waves that
// split the key range and merge their partials through LDS (single launch, no workspace).  Query rows of a kv head are the
// (token, q-head-in-group) pairs, token-major, so GQA shares every K/V byte across the group.
// Math per 32-key tile (all MFMA 16x16x32 bf16, fp32 accumulate):
//   S^T[key][row] = K . Q^T      (A = K rows straight from the paged cache, B = Q fragments in VGPRs)
//   online softmax in registers  (row statistics: 8 local values + 2 cross-lane steps)
//   O^T[d][row]  += V^T . P^T    (A = V^T via ds_read_b64_tr_b16 from an LDS-staged V tile,
//                                 B = P^T, which is exactly the S^T accumulator layout -> no shuffles)
// Split-KV partials (m, l, O) are merged by attn_combine_kernel in a fixed order.
#include "common.h"

struct AttnParams {
  const bf16_t* q;
  const bf16_t* kc;
  const bf16_t* vc;
  const int32_t* block_tables;
  const int32_t* context_lens;
  const int32_t* cu_q;
  const int32_t* tree_jidx;  // [B][tree_mq] glue position of each branch, or null -> branch / fan_out
  float* ws_o;               // [T*nh][splits][HD]
  float* ws_ml;              // [T*nh][splits][2]
  bf16_t* out_rows;          // [T][nh*HD] or null
  u32x2_t* out_frag;         // fragment-major [T][nh*HD] or null
  int max_blocks, q_per_seq, nh, nkv, bs;
  int mode;                  // 0 = causal (bottom-right aligned), 1 = tree
  int tree_K, tree_mq, tree_step, tree_F;
  int splits, use_tr, p_split;
  int bs_shift;              // log2(bs); block sizes are powers of two >= 16
  float scale_log2e;
  // attention + o_proj in one launch (attn_kernel<.., OPROJ = true>, ssd_attn_oproj_parts below)
  const u32x4_t* ow;         // o_proj weights, fragment-major [oN][nh*HD]
  float* oparts;             // fp32 slabs [nkv][Tq][oN]: slab h = W_o[:, columns of kv head h's q heads] . attention output of those heads
  int oN;
};

// LDS per wave: the V tile (32 keys x HD bf16) during the scan, then the wave's partial (O fp32 [RT*16][HD],
// m/l [RT*16][2]) for the in-block merge.
template <int HD, int RT>
constexpr int attn_region_bytes() { return RT * 16 * HD * 4 + RT * 16 * 8; }

// OPROJ (single sequence, <= 32 query rows per kv head, 8 waves, no grid key-splits): workgroup (kv head h, chunk c) of a
// grid of nkv * (oN / 128) first puts its slice of the o_proj weights in flight -- 8 row groups x the G*HD columns of kv head
// h's q heads, one row group per wave, <= 8 KiB per wave straight to VGPRs -- then computes the attention of kv head h exactly
// as below (the oN / 128 workgroups of a head repeat it: its K/V pages sit in one XCD's L2, workgroup id % nkv = h), leaves the
// normalised bf16 output of its heads in LDS as an MFMA B operand and multiplies: slab h of the split-K partial sums of
// o_proj, summed by the consumer like ssd_gemm_parts' slabs.  The weight stream hides behind the attention latency chain and
// one kernel boundary per layer disappears (a 1B draft decode layer: 6.6 us attention + 3.9 us o_proj as two launches).
template <int HD, int RT, int KT, bool OPROJ = false>
__global__ void __launch_bounds__(512) attn_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KTS = OPROJ ? 7 : 6;        // trace slot (profiling builds only)
  KTRACE(KTS, 0);
  constexpr int DS = HD / 32;  // k-steps of QK^T
  constexpr int DT = HD / 16;  // 16-wide output d tiles
  constexpr int REGION = attn_region_bytes<HD, RT>();
  const int lane = threadIdx.x & 63, r16 = lane & 15, g4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int W = blockDim.x >> 6;
  bf16_t* vlds = reinterpret_cast<bf16_t*>(smem + (size_t)wave * REGION);
  // grid = (B * nkv, row-tile groups, key splits): the (sequence, kv head) index is the FASTEST grid dimension, so with 8 kv
  // heads every workgroup of one kv head -- all the row tiles of a tree step / a prefill, all key splits -- is dispatched to the
  // same XCD (workgroup id % 8) and shares that head's K/V pages in ONE L2 instead of fetching them into eight
  const int b = OPROJ ? 0 : (int)(blockIdx.x / p.nkv), h = blockIdx.x % p.nkv;
  const int G = p.nh / p.nkv;
  u32x4_t wreg[OPROJ ? 8 : 1];
  const int q0 = p.cu_q ? p.cu_q[b] : b * p.q_per_seq;
  const int Tq = p.cu_q ? (p.cu_q[b + 1] - q0) : p.q_per_seq;
  const int rows = Tq * G;
  const int tile_base = blockIdx.y * RT;
  if (tile_base * 16 >= rows) return;   // block-uniform
  const int ctx = p.context_lens[b];
  const int32_t* bt = p.block_tables + (size_t)b * p.max_blocks;
  const int z = blockIdx.z;

  // ---- per-lane query-row descriptors and Q fragments ----
  u32x4_t qf[RT][DS];
  int tl[RT], hq[RT];
  bool rvalid[RT];
  int lim[RT];       // causal: keys [0, lim) visible.  tree: prefix length P
  int tj[RT];        // tree: glue position j of the branch
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int rho = (tile_base + rt) * 16 + r16;
    rvalid[rt] = rho < rows;
    const int rr = rvalid[rt] ? rho : 0;
    tl[rt] = rr / G;
    hq[rt] = h * G + (rr % G);
    const bf16_t* qp = p.q + ((size_t)(q0 + tl[rt]) * p.nh + hq[rt]) * HD + g4 * 8;
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) qf[rt][ds] = *reinterpret_cast<const u32x4_t*>(qp + ds * 32);
    if (p.mode == 0) {
      lim[rt] = ctx - (Tq - 1 - tl[rt]);
      tj[rt] = 0;
    } else {
      lim[rt] = ctx - (p.tree_K + 1) - (p.tree_step + 1) * p.tree_mq;
      tj[rt] = p.tree_jidx ? p.tree_jidx[b * p.tree_mq + tl[rt]] : (tl[rt] / p.tree_F);
    }
  }

  // ---- key tiles of this (grid split z, wave) pair: 32-key tiles dealt round-robin over all parts, so the
  // first tile's address needs nothing but the wave index (its page-table entry loads in parallel with ctx) ----
  int kmax = ctx;
  if (p.mode == 0) {  // rows of this tile never look past the last row's causal limit
    const int last_row = min(rows - 1, (tile_base + RT) * 16 - 1);
    kmax = ctx - (Tq - 1 - last_row / G);
  }
  const int stride = p.splits * W * 32;
  int k0 = (z * W + wave) * 32;

  float m[RT], lsum[RT];
  f32x4_t o[RT][DT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    m[rt] = -INFINITY; lsum[rt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[rt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  constexpr int CH = HD / 8;             // 16-byte chunks per key row
  constexpr int NV = (32 * CH) / 64;     // V chunks per lane per tile
  // page-table entries of a tile are wave-uniform (scalar loads): keys k..k+15 share a page and so do k+16..k+31
  // (block sizes are multiples of 16); a key clamped to ctx-1 falls into one of the two.  The index is clamped to
  // the table, not to ctx, so the lookup can be issued before ctx is known; entries past ctx are never used.
  auto blk = [&](int key) { return key >> p.bs_shift; };   // block sizes are powers of two (checked on the host)
  auto pages = [&](int kt, int& pa, int& pb) {
    pa = bt[min(blk(kt), p.max_blocks - 1)];
    pb = bt[min(blk(kt + 16), p.max_blocks - 1)];
  };
  // Addresses = a wave-uniform 64-bit base per (tile, 16-key half) + a 32-bit per-lane offset.  Rows past ctx-1 are
  // clamped to the last valid row of the tile (never an unwritten row, never an unallocated page): if the second half
  // starts at or past ctx it is redirected to the first half's page.
  auto half_base = [&](const bf16_t* base, int page, int key0) -> const char* {
    return reinterpret_cast<const char*>(base + (((size_t)page * p.nkv + h) * p.bs + (key0 & (p.bs - 1))) * HD);
  };
  auto issue_k = [&](int kt, int pa, int pb, u32x4_t (&kf)[2][DS]) {
    const int cl = ctx - 1 - kt;                  // >= 0 for every tile we touch
    const bool bvalid = cl >= 16;
    const char* b0 = half_base(p.kc, pa, kt);
    const char* b1 = bvalid ? half_base(p.kc, pb, kt + 16) : b0;
    const uint32_t o0 = (uint32_t)(min(r16, cl) * HD + g4 * 8) * 2u;
    const uint32_t o1 = (uint32_t)(min(r16, bvalid ? cl - 16 : cl) * HD + g4 * 8) * 2u;
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) {
      kf[0][ds] = *reinterpret_cast<const u32x4_t*>(b0 + o0 + ds * 64);
      kf[1][ds] = *reinterpret_cast<const u32x4_t*>(b1 + o1 + ds * 64);
    }
  };
  auto issue_v = [&](int kt, int pa, int pb, u32x4_t (&vf)[NV]) {
    const int cl = ctx - 1 - kt;
    const bool bvalid = cl >= 16;
    const char* b0 = half_base(p.vc, pa, kt);
    const char* b1 = bvalid ? half_base(p.vc, pb, kt + 16) : b0;
    const int clb = bvalid ? cl - 16 : cl;
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      constexpr int KPI = 64 / CH;                // keys covered by one 64-lane pass
      const int key = it * KPI + lane / CH;       // 0..31; the half is static per `it`
      const bool second = (it * KPI) >= 16;
      const int r = second ? min(key - 16, clb) : min(key, cl);
      const uint32_t off = (uint32_t)(r * HD + (lane % CH) * 8) * 2u;
      vf[it] = *reinterpret_cast<const u32x4_t*>((second ? b1 : b0) + off);
    }
  };

  // KT tiles are in flight per wave: their K/V loads are issued together and a slot's registers are refilled with
  // the tile KT ahead as soon as they have been consumed (the loads fly during the softmax and P.V of this and the
  // following tiles); the page entries for that refill were fetched (scalar loads) one round earlier.
  constexpr bool PIPE_K = !(HD == 128 && RT == 2);   // the widest variant has no registers to hold K ahead of time
  u32x4_t kreg[KT][2][DS], vreg[KT][NV];
  int ca[KT], cb[KT], na[KT], nb[KT];
#pragma unroll
  for (int j = 0; j < KT; ++j) {
    pages(k0 + j * stride, ca[j], cb[j]);
    pages(k0 + (j + KT) * stride, na[j], nb[j]);
  }
#pragma unroll
  for (int j = 0; j < KT; ++j)
    if (k0 + j * stride < kmax) {
      if constexpr (PIPE_K) issue_k(k0 + j * stride, ca[j], cb[j], kreg[j]);
      issue_v(k0 + j * stride, ca[j], cb[j], vreg[j]);
    }
  if constexpr (OPROJ) {
    // the o_proj slice is needed LAST (after the merge); until round 3 its 8 KiB per wave were the first loads of the kernel,
    // i.e. the oldest in the in-order return queue: every wait of the attention proper -- Q, the first K / V tiles -- was a wait
    // for an HBM round trip of weights as well (first K / V issue at +4.6 us, profiles/r04_ktrace_1b_before.txt).  Issued here,
    // behind the first tiles, they stream while the keys are processed
    const int KTo = (p.nh * HD) >> 5, KS = (G * HD) >> 5;                   // k-tiles of o_proj's K, of one kv head's slice (<= 8)
    const u32x4_t* wp = p.ow + ((size_t)(((blockIdx.x / p.nkv) * 8 + wave) * KTo + h * KS) << 6) + lane;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) wreg[kt] = __builtin_nontemporal_load(wp + ((size_t)min(kt, KS - 1) << 6));   // clamped: unconditional loads
  }
  KTRACE(KTS, 1);

  for (; k0 < kmax; k0 += KT * stride) {
#pragma unroll
   for (int j = 0; j < KT; ++j) {
    const int kt = k0 + j * stride;        // this tile
    if (kt >= kmax) break;                 // wave-uniform
    const int kn = kt + KT * stride;       // the tile that refills slot j
    const bool more = kn < kmax;
    if constexpr (!PIPE_K) issue_k(kt, ca[j], cb[j], kreg[j]);
    // -- stage the V tile [32 keys][HD] into this wave's LDS region as DT sub-tiles of [32 keys][16 d] --
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int c = it * 64 + lane;
      const int key = c / CH, d8 = c % CH;
      *reinterpret_cast<u32x4_t*>(vlds + ((d8 >> 1) * 32 + key) * 16 + (d8 & 1) * 8) = vreg[j][it];
    }
    if (more) issue_v(kn, na[j], nb[j], vreg[j]);
    // -- S^T = K . Q^T --
    f32x4_t st[2][RT];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        st[hf][rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) st[hf][rt] = mfma16(kreg[j][hf][ds], qf[rt][ds], st[hf][rt]);
      }
    }
    if constexpr (PIPE_K) { if (more) issue_k(kn, na[j], nb[j], kreg[j]); }
    ca[j] = na[j]; cb[j] = nb[j];
    pages(kn + KT * stride, na[j], nb[j]);
    const int k0t = kt;
    // -- mask + online softmax; P^T stays in the accumulator layout --
    u32x4_t pf[RT], pl[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float s[8];
      float tmax = -INFINITY;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = k0t + hf * 16 + g4 * 4 + r;
          bool ok = rvalid[rt] && key < kmax;
          if (p.mode == 0) {
            ok = ok && key < lim[rt];
          } else {
            const int P = lim[rt];
            const int rel = key - P;
            bool vis = rel < 0;                                            // trunk prefix
            vis = vis || (rel <= p.tree_K && rel <= tj[rt]);                // glue columns 0..j
            vis = vis || (rel > p.tree_K && ((rel - p.tree_K - 1) % p.tree_mq) == tl[rt]);  // own branch diagonal
            ok = ok && vis;
          }
          const float v = ok ? st[hf][rt][r] * p.scale_log2e : -INFINITY;
          s[hf * 4 + r] = v;
          tmax = fmaxf(tmax, v);
        }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float mnew = fmaxf(m[rt], tmax);
      const float msafe = (mnew == -INFINITY) ? 0.f : mnew;
      const float alpha = exp2f(m[rt] - msafe);
      float ps = 0.f;
      float pv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { pv[i] = exp2f(s[i] - msafe); ps += pv[i]; }
      lsum[rt] = lsum[rt] * alpha + ps;
      m[rt] = mnew;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) o[rt][dt] *= alpha;
      pf[rt] = u32x4_t{pack_bf2(pv[0], pv[1]), pack_bf2(pv[2], pv[3]), pack_bf2(pv[4], pv[5]), pack_bf2(pv[6], pv[7])};
      // P = P_hi + P_lo (two bf16 terms, ~2^-17 relative): P.V then tracks an fp32-softmax reference instead of
      // carrying FlashAttention's 2^-9 probability rounding; costs one extra MFMA per output tile.
      float pr[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) pr[i] = pv[i] - round_bf(pv[i]);
      pl[rt] = u32x4_t{pack_bf2(pr[0], pr[1]), pack_bf2(pr[2], pr[3]), pack_bf2(pr[4], pr[5]), pack_bf2(pr[6], pr[7])};
    }
    // the V tile is private to this wave: LDS ops of one wave execute in order, so a wave-level fence (no
    // s_barrier -- waves have different trip counts) is all the write->transpose-read hand-off needs
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // -- O^T += V^T . P^T ; V^T fragments by transpose-read from the staged tile --
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      u32x4_t vf;
      if (p.use_tr) {
        // lane (r16, g4) supplies the address of row 4*g4 + (r16 >> 2), column quad (r16 & 3) of the
        // [32 keys][16 d] sub-tile; the hardware returns column r16 of rows 4*g4 .. 4*g4+3.
        const bf16_t* a0 = vlds + (dt * 32 + g4 * 4 + (r16 >> 2)) * 16 + (r16 & 3) * 4;
        typedef s16x4_t __attribute__((address_space(3))) * lds_v4_t;
        const s16x4_t t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(a0));
        const s16x4_t t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(a0 + 16 * 16));
        u32x2_t lo = __builtin_bit_cast(u32x2_t, t0), hi = __builtin_bit_cast(u32x2_t, t1);
        vf = u32x4_t{lo[0], lo[1], hi[0], hi[1]};
      } else {
        uint32_t e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int key = (i < 4) ? (g4 * 4 + i) : (16 + g4 * 4 + (i - 4));
          e[i] = vlds[(dt * 32 + key) * 16 + r16];
        }
        vf = u32x4_t{e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16)};
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        o[rt][dt] = mfma16(vf, pf[rt], o[rt][dt]);
        if (p.p_split) o[rt][dt] = mfma16(vf, pl[rt], o[rt][dt]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
   }
  }

  // ---- epilogue ----
  KTRACE(KTS, 2);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    lsum[rt] += __shfl_xor(lsum[rt], 16, 64);
    lsum[rt] += __shfl_xor(lsum[rt], 32, 64);
  }
  if (W == 1) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      if (!rvalid[rt]) continue;
      const float l = lsum[rt];
      const size_t row = (size_t)(q0 + tl[rt]) * p.nh + hq[rt];
      if (p.splits == 1) {
        const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int d = dt * 16 + g4 * 4;
          const u32x2_t v = {pack_bf2(o[rt][dt][0] * inv, o[rt][dt][1] * inv), pack_bf2(o[rt][dt][2] * inv, o[rt][dt][3] * inv)};
          if (p.out_rows) *reinterpret_cast<u32x2_t*>(p.out_rows + row * HD + d) = v;
          if (p.out_frag) {
            const int kcol = hq[rt] * HD + d;
            p.out_frag[frag_chunk(q0 + tl[rt], kcol >> 3, (p.nh * HD) >> 5) * 2 + ((kcol >> 2) & 1)] = v;
          }
        }
      } else {
        float* wo = p.ws_o + (row * p.splits + z) * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4_t*>(wo + dt * 16 + g4 * 4) = o[rt][dt];
        if (g4 == 0) {
          p.ws_ml[(row * p.splits + z) * 2] = m[rt];
          p.ws_ml[(row * p.splits + z) * 2 + 1] = l;
        }
      }
    }
    return;
  }
  // ---- W > 1: merge the waves' partials through LDS in wave order (deterministic), one launch, no workspace ----
  {
    float* po = reinterpret_cast<float*>(smem + (size_t)wave * REGION);
    float* pml = po + RT * 16 * HD;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int lr = rt * 16 + r16;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4_t*>(po + lr * HD + dt * 16 + g4 * 4) = o[rt][dt];
      if (g4 == 0) { pml[lr * 2] = m[rt]; pml[lr * 2 + 1] = lsum[rt]; }
    }
  }
  __syncthreads();
  KTRACE(KTS, 3);
  for (int item = threadIdx.x; item < RT * 16 * (HD / 4); item += blockDim.x) {
    const int lr = item / (HD / 4), d = (item % (HD / 4)) * 4;
    const int rho = tile_base * 16 + lr;
    if (rho >= rows) continue;
    // all of the (<= 8) waves' partials are read first, then merged in wave order: the rolled loops were ds_read -> wait -> use
    // per wave, twice (1.3-1.5 us of merge in a 10-13 us kernel, profiles/r04_ktrace_1b_before.txt); slots past W re-read the last
    // wave's and are not used -- same operations in the same order, bit-identical
    float mw[8], lw[8];
    f32x4_t ow[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const float* base = reinterpret_cast<const float*>(smem + (size_t)min(w, W - 1) * REGION);
      mw[w] = base[RT * 16 * HD + lr * 2];
      lw[w] = base[RT * 16 * HD + lr * 2 + 1];
      ow[w] = *reinterpret_cast<const f32x4_t*>(base + lr * HD + d);
    }
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < 8; ++w)
      if (w < W) M = fmaxf(M, mw[w]);
    const float Ms = (M == -INFINITY) ? 0.f : M;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    float L = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w)
      if (w < W) {
        const float wt = exp2f(mw[w] - Ms);
        L += wt * lw[w];
        acc += wt * ow[w];
      }
    const int tok = q0 + rho / G, head = h * G + rho % G;
    const size_t row = (size_t)tok * p.nh + head;
    if constexpr (OPROJ) {
      // x[m = token][k = (q head within the group) * HD + d] as the B operand of the o_proj product: 16-byte chunk
      // (k / 8, m), fragment-major within this kv head's K slice
      const float inv = L > 0.f ? 1.0f / L : 0.f;
      const u32x2_t v = {pack_bf2(acc[0] * inv, acc[1] * inv), pack_bf2(acc[2] * inv, acc[3] * inv)};
      const int kl = (rho % G) * HD + d;
      reinterpret_cast<u32x2_t*>(smem + (size_t)W * REGION)[(((kl >> 3) << 4) + rho / G) * 2 + ((kl >> 2) & 1)] = v;
    } else if (p.splits == 1) {
      const float inv = L > 0.f ? 1.0f / L : 0.f;
      const u32x2_t v = {pack_bf2(acc[0] * inv, acc[1] * inv), pack_bf2(acc[2] * inv, acc[3] * inv)};
      if (p.out_rows) *reinterpret_cast<u32x2_t*>(p.out_rows + row * HD + d) = v;
      if (p.out_frag) {
        const int kcol = head * HD + d;
        p.out_frag[frag_chunk(tok, kcol >> 3, (p.nh * HD) >> 5) * 2 + ((kcol >> 2) & 1)] = v;
      }
    } else {
      *reinterpret_cast<f32x4_t*>(p.ws_o + (row * p.splits + z) * HD + d) = acc;
      if (d == 0) {
        p.ws_ml[(row * p.splits + z) * 2] = M;
        p.ws_ml[(row * p.splits + z) * 2 + 1] = L;
      }
    }
  }
  KTRACE(KTS, 4);
  if constexpr (OPROJ) {
    __syncthreads();
    // slab h of o_proj: this wave's row group x the K slice of kv head h, all k-tiles already in registers
    const int KS = (G * HD) >> 5;
    const u32x4_t* xl = reinterpret_cast<const u32x4_t*>(smem + (size_t)W * REGION);
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      if (kt < KS) {          // block-uniform
        u32x4_t xb = {0u, 0u, 0u, 0u};
        if (r16 < Tq) xb = xl[((kt * 4 + g4) << 4) + r16];
        acc = mfma16(wreg[kt], xb, acc);
      }
    }
    if (r16 < Tq) {
      const int n = ((blockIdx.x / p.nkv) * 8 + wave) * 16 + g4 * 4;
      *reinterpret_cast<f32x4_t*>(p.oparts + ((size_t)h * Tq + r16) * p.oN + n) = acc;
    }
    KTRACE(KTS, 5);
  }
}

// Merge split-KV partials: out = sum_s 2^(m_s - M) O_s / sum_s 2^(m_s - M) l_s, splits visited in order.
template <int HD>
__global__ void attn_combine_kernel(const float* __restrict__ ws_o, const float* __restrict__ ws_ml, int splits,
                                    int nh, bf16_t* __restrict__ out_rows, u32x2_t* __restrict__ out_frag) {
  const int row = blockIdx.x;  // token * nh + head
  const int d = threadIdx.x * 4;
  float M = -INFINITY;
  for (int s = 0; s < splits; ++s) M = fmaxf(M, ws_ml[((size_t)row * splits + s) * 2]);
  const float Ms = (M == -INFINITY) ? 0.f : M;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  float L = 0.f;
  for (int s = 0; s < splits; ++s) {
    const float w = exp2f(ws_ml[((size_t)row * splits + s) * 2] - Ms);
    L += w * ws_ml[((size_t)row * splits + s) * 2 + 1];
    acc += w * *reinterpret_cast<const f32x4_t*>(ws_o + ((size_t)row * splits + s) * HD + d);
  }
  const float inv = L > 0.f ? 1.0f / L : 0.f;
  const u32x2_t v = {pack_bf2(acc[0] * inv, acc[1] * inv), pack_bf2(acc[2] * inv, acc[3] * inv)};
  if (out_rows) *reinterpret_cast<u32x2_t*>(out_rows + (size_t)row * HD + d) = v;
  if (out_frag) {
    const int token = row / nh, head = row % nh;
    const int kcol = head * HD + d;
    out_frag[frag_chunk(token, kcol >> 3, (nh * HD) >> 5) * 2 + ((kcol >> 2) & 1)] = v;
  }
}

template <int HD, int RT, int KT>
static int attn_launch_rt(const AttnParams& p, dim3 grid, int waves, hipStream_t st) {
  const int lds = waves * attn_region_bytes<HD, RT>();
  auto kern = attn_kernel<HD, RT, KT, false>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
    return SSD_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, grid, dim3(64 * waves), lds, st, p);
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

// Row tiles per workgroup: two for prefill-sized query blocks (> 8 tiles; K/V bytes shared by 32 rows), one for the decode-side
// shapes (more workgroups: the scan is bound by per-CU load bandwidth and latency, not by K/V bytes).
template <int HD>
static int attn_launch(const AttnParams& p, int B, int T, int max_q, int waves, int force_rt1, hipStream_t st) {
  const int G = p.nh / p.nkv;
  const int row_tiles = (max_q * G + 15) / 16;
  // (the 24-branch tree step has 6 row tiles per kv head: one tile per workgroup measured 10.5 -> 7.7 us at ctx 150 and
  // 19.6 -> 12.6 us at ctx 640 on the 1B draft, profiles/r02_draft_probe.txt)
  const int rt = (row_tiles > 8 && !force_rt1) ? 2 : 1;
  dim3 grid(B * p.nkv, (row_tiles + rt - 1) / rt, p.splits);
  // one key tile in flight per wave (KT = 1): measured on MI355X, KT = 2/4 buy nothing -- with 8 waves per workgroup
  // the scan is bound by the handful of CUs it occupies, not by a single wave's load latency
  const int rc = rt == 2 ? attn_launch_rt<HD, 2, 1>(p, grid, waves, st) : attn_launch_rt<HD, 1, 1>(p, grid, waves, st);
  if (rc != SSD_OK) return rc;
  if (p.splits > 1) {
    hipLaunchKernelGGL((attn_combine_kernel<HD>), dim3(T * p.nh), dim3(HD / 4), 0, st, p.ws_o, p.ws_ml, p.splits, p.nh,
                       p.out_rows, p.out_frag);
    if (hipGetLastError() != hipSuccess) return SSD_ERR_LAUNCH;
  }
  return SSD_OK;
}

// Attention of ONE sequence's T query rows (causal, bottom-right aligned: single-token decode or a K+1-row glue / verify) fused
// with o_proj: parts receives nkv fp32 slabs [nkv][T][N] whose sum over the slabs is o_proj(attention output) -- consumed like
// ssd_gemm_parts' slabs (ssd_gemm_fused_parts / ssd_rmsnorm_parts with splits = nkv).  Replaces ssd_attn_paged + ssd_gemm_parts
// (reference ssd/layers/attention.py:105-111,126-131 + ssd/layers/linear.py:186-199 o_proj) for the single-GPU drafts.
// Constraints: T * (nh / nkv) <= 32 (hd 64) or <= 16 (hd 128); (nh / nkv) * hd <= 256; N % 128 == 0; nkv <= 16; context
// within one launch (no grid key-splits: use it for context buckets <= 1024).
extern "C" int ssd_attn_oproj_parts(const void* q_rows, const void* k_cache, const void* v_cache, const int32_t* block_tables,
                                    int max_blocks, const int32_t* context_lens, int T, int nh, int nkv, int hd, int block_size,
                                    float scale, const void* w_o_frag, int N, void* parts, void* stream) {
  if (T <= 0 || nh <= 0 || nkv <= 0 || nh % nkv || nkv > 16) return SSD_ERR_SHAPE;
  if (hd != 64 && hd != 128) return SSD_ERR_SHAPE;
  const int G = nh / nkv, rows = T * G;
  if (rows > (hd == 64 ? 32 : 16) || T > 16 || G * hd > 256 || ((G * hd) & 31) || N <= 0 || (N & 127)) return SSD_ERR_SHAPE;
  if (block_size < 16 || (block_size & (block_size - 1)) != 0 || max_blocks <= 0) return SSD_ERR_SHAPE;
  if (!w_o_frag || !parts) return SSD_ERR_ARG;
  AttnParams p;
  p.q = (const bf16_t*)q_rows; p.kc = (const bf16_t*)k_cache; p.vc = (const bf16_t*)v_cache;
  p.block_tables = block_tables; p.context_lens = context_lens; p.cu_q = nullptr; p.tree_jidx = nullptr;
  p.ws_o = nullptr; p.ws_ml = nullptr; p.out_rows = nullptr; p.out_frag = nullptr;
  p.max_blocks = max_blocks; p.q_per_seq = T; p.nh = nh; p.nkv = nkv; p.bs = block_size;
  p.bs_shift = __builtin_ctz(block_size);
  p.mode = 0; p.tree_K = 0; p.tree_mq = 1; p.tree_step = 0; p.tree_F = 1;
  p.splits = 1; p.use_tr = 1; p.p_split = 1;
  p.scale_log2e = scale * 1.4426950408889634f;
  p.ow = (const u32x4_t*)w_o_frag; p.oparts = (float*)parts; p.oN = N;
  const dim3 grid(nkv * (N / 128), 1, 1), block(512);
  hipStream_t st = (hipStream_t)stream;
  const int rt = rows > 16 ? 2 : 1;
#define AO_LAUNCH(HDV, RTV)                                                                                          \
  do {                                                                                                               \
    const int lds = 8 * attn_region_bytes<HDV, RTV>() + 16 * 256 * 2;                                               \
    auto kern = attn_kernel<HDV, RTV, 1, true>;                                                                      \
    if (lds > 64 * 1024 &&                                                                                           \
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) \
      return SSD_ERR_LAUNCH;                                                                                         \
    hipLaunchKernelGGL(kern, grid, block, lds, st, p);                                                               \
  } while (0)
  if (hd == 64) { if (rt == 2) AO_LAUNCH(64, 2); else AO_LAUNCH(64, 1); }
  else AO_LAUNCH(128, 1);
#undef AO_LAUNCH
  return hipGetLastError() == hipSuccess ? SSD_OK : SSD_ERR_LAUNCH;
}

extern "C" int ssd_attn_paged(const void* q_rows, const void* k_cache, const void* v_cache,
                              const int32_t* block_tables, int max_blocks, const int32_t* context_lens,
                              const int32_t* cu_q, int q_per_seq, int B, int T, int max_q, int nh, int nkv, int hd,
                              int block_size, float scale, int mode, int tree_K, int tree_mq, int tree_step,
                              int tree_F, const int32_t* tree_jidx, int splits, int flags, void* ws_o, void* ws_ml,
                              void* out_rows, void* out_frag, void* stream) {
  if (B <= 0 || T <= 0 || max_q <= 0 || nh <= 0 || nkv <= 0 || nh % nkv) return SSD_ERR_SHAPE;
  if (hd != 64 && hd != 128) return SSD_ERR_SHAPE;
  if (block_size < 16 || (block_size & (block_size - 1)) != 0 || max_blocks <= 0) return SSD_ERR_SHAPE;
  if (((nh * hd) & 31) != 0) return SSD_ERR_SHAPE;
  if (splits < 1) return SSD_ERR_ARG;
  if (splits > 1 && (!ws_o || !ws_ml)) return SSD_ERR_ARG;
  if (mode == 1 && (tree_mq <= 0 || tree_F <= 0)) return SSD_ERR_ARG;
  AttnParams p;
  p.q = (const bf16_t*)q_rows; p.kc = (const bf16_t*)k_cache; p.vc = (const bf16_t*)v_cache;
  p.block_tables = block_tables; p.context_lens = context_lens; p.cu_q = cu_q; p.tree_jidx = tree_jidx;
  p.ws_o = (float*)ws_o; p.ws_ml = (float*)ws_ml; p.out_rows = (bf16_t*)out_rows; p.out_frag = (u32x2_t*)out_frag;
  p.max_blocks = max_blocks; p.q_per_seq = q_per_seq; p.nh = nh; p.nkv = nkv; p.bs = block_size;
  p.bs_shift = __builtin_ctz(block_size);
  p.mode = mode; p.tree_K = tree_K; p.tree_mq = tree_mq; p.tree_step = tree_step; p.tree_F = tree_F;
  p.splits = splits; p.use_tr = (flags & 1) ? 0 : 1; p.p_split = (flags & 2) ? 0 : 1;
  p.scale_log2e = scale * 1.4426950408889634f;
  p.ow = nullptr; p.oparts = nullptr; p.oN = 0;
  hipStream_t st = (hipStream_t)stream;
  // flags bits 8..11: waves per workgroup that split the key range and merge in LDS (default 1; 1..8)
  int waves = (flags >> 8) & 0xf;
  if (waves < 1) waves = 1;
  if (waves > 8) waves = 8;
  // flags bit 2: one 16-row tile per workgroup even for wider query blocks (twice the workgroups for the 24-branch tree)
  const int rt1 = (flags >> 2) & 1;
  return hd == 128 ? attn_launch<128>(p, B, T, max_q, waves, rt1, st) : attn_launch<64>(p, B, T, max_q, waves, rt1, st);
}

// The reference's two other attention call sites under their own names (SURVEY section 8b's list): thin forms of ssd_attn_paged, which already
// covers them as modes.
//   ssd_attn_prefill_varlen -- flash_attn_varlen_func, ssd/layers/attention.py:90-93: causal attention of B packed sequences (cu_q int32 [B + 1])
//                              over the paged cache the same forward has just filled (context_lens = the sequences' total lengths).
//   ssd_attn_tree           -- the flashinfer custom-mask prefill of the draft tree, ssd/layers/attention.py:113-125 + mask_helpers.py:12-21:
//                              tree_mq branch rows per sequence at tree step `tree_step`, structural mask (no mask tensor, no plan()).
extern "C" int ssd_attn_prefill_varlen(const void* q_rows, const void* k_cache, const void* v_cache, const int32_t* block_tables, int max_blocks,
                                       const int32_t* context_lens, const int32_t* cu_q, int B, int T, int max_q, int nh, int nkv, int hd,
                                       int block_size, float scale, void* out_rows, void* out_frag, void* stream) {
  if (!cu_q) return SSD_ERR_ARG;
  return ssd_attn_paged(q_rows, k_cache, v_cache, block_tables, max_blocks, context_lens, cu_q, 0, B, T, max_q, nh, nkv, hd, block_size, scale, 0, 0,
                        0, 0, 1, nullptr, 1, 0, nullptr, nullptr, out_rows, out_frag, stream);
}
extern "C" int ssd_attn_tree(const void* q_rows, const void* k_cache, const void* v_cache, const int32_t* block_tables, int max_blocks,
                             const int32_t* context_lens, int B, int tree_K, int tree_mq, int tree_step, int tree_F, const int32_t* tree_jidx,
                             int nh, int nkv, int hd, int block_size, float scale, void* out_rows, void* out_frag, void* stream) {
  if (tree_mq <= 0 || tree_K <= 0 || tree_step < 0) return SSD_ERR_SHAPE;
  return ssd_attn_paged(q_rows, k_cache, v_cache, block_tables, max_blocks, context_lens, nullptr, tree_mq, B, B * tree_mq, tree_mq, nh, nkv, hd,
                        block_size, scale, 1, tree_K, tree_mq, tree_step, tree_F, tree_jidx, 1, 0, nullptr, nullptr, out_rows, out_frag, stream);
}

KT_DEFINE_SETTER(attention)
