// Incremental decoder-step kernels for gfx950 (fp32) — SURVEY §8 rows A13-A19.
//
// The reference re-runs the whole prefix every step (O(T²)); its decoder is strictly causal, so we
// process ONE new position per step against per-position caches.  Beam re-ordering never moves a
// cache: `anc[n][j]` names the slot (sequence index at the time position j was processed) whose
// entry belongs to sequence n's history.  Every kernel reads the current position from device
// memory (`*pos`) so the same captured graph serves every step.
//
//   dec_embed_kernel         y = embed[tok]·sqrt(d) + pos_table[pos]
//   dynexp_step_kernel       DynamicExpansionBlock for the newest row (layers.py:152-204)
//   cross_attn_step_kernel   MultiHeadAttention against per-image cached K/V (layers.py:266-295)
//   logsoftmax_topk_kernel   log_softmax over the vocabulary + top-k (captioning_model.py:162-170)
//   beam_step_kernel         candidate masking, k·k selection, prefix/ancestor re-gather (:172-223)
//   beam_finalize_kernel     length-normalised ranking (:225-227)
#include "odic_common.h"

namespace {

constexpr int MAX_T = 128;     // max decode positions supported by the LDS scratch below
constexpr int MAX_E = 32;      // max expansion vectors per position
constexpr int MAX_K = 16;      // max beam size

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_embed_kernel(const long long* __restrict__ tokens,
                                                        const float* __restrict__ embed,
                                                        const float* __restrict__ pos_table,
                                                        const int* __restrict__ pos, float* __restrict__ y,
                                                        long ldy, int N, int d, int pos_rows, float scale) {
  const int n = blockIdx.x;
  const long tok = tokens[n];
  const int p = *pos;
  if (p < 0 || p >= pos_rows) return;   // *pos is device memory no host check can see: a replay past the table's last
                                        // row (End_ExpansionNet_v2.py:105 caps the positions at max_seq_len) writes nothing
  for (int c = threadIdx.x; c < d; c += blockDim.x)
    y[n * ldy + c] = embed[tok * d + c] * scale + pos_table[(long)p * d + c];
}

// ---------------------------------------------------------------------------------------------
// Dynamic expansion for the newest position.  With t = *pos and slot(j) = (j < t ? anc[n][j] : n):
//   forward  z_fw[e][i] = (qexp[e] + cond_t)·key_i / sqrt(d)            i <= t
//            wfa_t[i][e] = relu(z_fw)/(Σ_i relu(z_fw) + eps),  wfb_t with -z        (CACHED: (t+1)·E scalars each)
//            class vectors of the reference:  afull_t[e] = Σ_i wfa_t[i][e]·va_i + bexp[e] + cond_t   (never formed)
//   backward z_bw[j][e] = (qexp[e] + cond_j)·key_t / sqrt(d)            j <= t
//            wba[j][e] = relu(z_bw)/(Σ_{j,e} relu(z_bw) + eps),  wbb with -z
//            out_a = Σ_{j,e} wba[j][e]·afull_j[e]
//                  = Σ_i ca[i]·va_i + Σ_e wea[e]·bexp[e] + Σ_j wja[j]·cond_j
//              with ca[i] = Σ_{j>=i} Σ_e wba[j][e]·wfa_j[i][e],  wea[e] = Σ_j wba[j][e],  wja[j] = Σ_e wba[j][e]
//   y_out = y_in + σ(sel)·out_a + (1-σ(sel))·out_b        (nothing is added on a padded row)
// The reference (layers.py:152-204) materialises the (t·E) x d class matrices every step; a cache of afull / bfull
// per position would be 2·E·d floats per position and sequence (64 KB at E = 16, d = 512: 100 MB read per step at
// t = 10 for 48 sequences x 3 layers, streamed beside the encoder's GEMMs).  Re-associating the double sum moves the
// cache to the forward WEIGHTS — (t+1)·E scalars per position — and the step to 3·(t+1) + E rows of d floats per
// sequence: 8x fewer bytes, same arithmetic up to summation order.
// The dot products split as qexp[e]·key + cond·key; qexp[e]·key_j is cached per position (qk_c).
// ---------------------------------------------------------------------------------------------
struct DynParams {
  const float* lin; long ldlin; const float* qexp; const float* bexp;
  float* cond_c; float* key_c; float* va_c; float* vb_c; float* wfa_c; float* wfb_c; float* qk_c;
  const int* anc; const int* row_valid; const int* pos; const float* y_in; long ldyi; float* y; long ldy;
  int N, T, d, E; float eps;
};

// One block per sequence, 1024 threads.  Phase 1: cache writes, the 2t+E+1 dot products (one per 16-lane group,
// 32 in flight per block, float4 loads), the normalised forward / backward weights, and the coefficients of
// the re-associated sum → LDS.  Phase 2 (two threads per channel): the sums over the cached value, condition and
// bias rows — (3·(t+1) + E) / 2 independent, coalesced loads per thread.
#ifndef ODIC_DYN_NT
#define ODIC_DYN_NT 1024
#endif
constexpr int DYN_NT = ODIC_DYN_NT;         // threads per sequence: the step is a chain of dependent load rounds, so the
                                     // block is as wide as it can be (64 dot-product groups, 2 x 512 channel threads)
__global__ __launch_bounds__(DYN_NT) void dynexp_step_kernel(DynParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int NTD = DYN_NT;
  const int d = p.d, E = p.E, T = p.T, n = blockIdx.x, tid = threadIdx.x;
  const int t = *p.pos;
  if (t < 0 || t >= T) return;         // the caches hold T positions: a step replayed past the last one changes nothing
  float* cond_t = sm;                  // [d]
  float* key_t = cond_t + d;           // [d]
  float* dk = key_t + d;               // [T]  cond_t·key_j
  float* ck = dk + T;                  // [T]  cond_j·key_t
  float* qk_t = ck + T;                // [MAX_E]  qexp[e]·key_t
  float* nfw = qk_t + MAX_E;           // [2*MAX_E] 1/(Σ_j relu(±z_fw[e][·]) + eps)
  float* red = nfw + 2 * MAX_E;        // [16]
  int* slot = (int*)(red + 16);        // [T]
  float* qkh = (float*)(slot + T);     // [T*E]  qexp[e]·key_j history
  float* wba = qkh + T * E;            // [T*E]  backward weights of this step
  float* wbb = wba + T * E;            // [T*E]
  float* scr = wbb + T * E;            // ca[T] | cb[T] | wja[T] | wjb[T] | wea[E] | web[E]
  float* comb = scr + 4 * T + 2 * E;   // [NTD]  second half's partial sums of phase 2

  const float* lin = p.lin + (long)n * p.ldlin;
  const float inv_sqrt_d = rsqrtf((float)d);
  const long NT = (long)p.N;
  const int TE = T * E;

  for (int c = tid; c < d; c += NTD) {
    const float cv = lin[c], kv = lin[d + c];
    cond_t[c] = cv; key_t[c] = kv;
    const long o = ((long)t * NT + n) * d + c;
    p.cond_c[o] = cv; p.key_c[o] = kv; p.va_c[o] = lin[2 * d + c]; p.vb_c[o] = lin[3 * d + c];
  }
  for (int j = tid; j <= t; j += NTD) {
    const int sl = j < t ? p.anc[(long)n * T + j] : n;
    slot[j] = sl;
  }
  __syncthreads();
  for (int i = tid; i < t * E; i += NTD) {
    const int j = i / E, e = i - j * E;
    qkh[i] = p.qk_c[((long)j * NT + slot[j]) * E + e];
  }

  // dot products, one per 16-lane group (64 groups):
  //   items 0..E-1: qk_t[e];  E..E+t: dk[j] (j = 0..t);  E+t+1 .. E+2t: ck[j] (j = 0..t-1)
  const int nitems = E + (t + 1) + t;
  const int grp = tid >> 4, gl = tid & 15;
  for (int it = grp; it < nitems; it += NTD / 16) {
    const float* a; const float* b;
    if (it < E) { a = p.qexp + (long)it * d; b = key_t; }
    else if (it < E + t + 1) {
      const int j = it - E;
      a = cond_t; b = j < t ? p.key_c + ((long)j * NT + slot[j]) * d : key_t;
    } else {
      const int j = it - E - t - 1;
      a = p.cond_c + ((long)j * NT + slot[j]) * d; b = key_t;
    }
    float s = 0.f;
#pragma unroll 8
    for (int c = gl * 4; c < d; c += 64) {
      const float4 av = *(const float4*)(a + c);
      const float4 bv = *(const float4*)(b + c);
      s = fmaf(av.x, bv.x, s); s = fmaf(av.y, bv.y, s); s = fmaf(av.z, bv.z, s); s = fmaf(av.w, bv.w, s);
    }
    s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
    if (gl == 0) {
      if (it < E) { qk_t[it] = s; p.qk_c[((long)t * NT + n) * E + it] = s; }
      else if (it < E + t + 1) dk[it - E] = s;
      else ck[it - E - t - 1] = s;
    }
  }
  __syncthreads();
  if (tid == 0) ck[t] = dk[t];           // cond_t·key_t
  __syncthreads();

  // forward normalisers: 32 lanes per expansion query (E <= 32 → all of them in one pass), fixed shuffle order
  {
    const int e = tid >> 5, q = tid & 31;
    if (e < E) {
      float sp = 0.f, sn = 0.f;
      for (int j = q; j <= t; j += 32) {
        const float qq = j < t ? qkh[j * E + e] : qk_t[e];
        const float z = (qq + dk[j]) * inv_sqrt_d;
        sp += fmaxf(z, 0.f); sn += fmaxf(-z, 0.f);
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) { sp += __shfl_xor(sp, o, 64); sn += __shfl_xor(sn, o, 64); }
      if (q == 0) { nfw[e] = 1.0f / (sp + p.eps); nfw[MAX_E + e] = 1.0f / (sn + p.eps); }
    }
  }
  float bp = 0.f, bn = 0.f;
  for (int i = tid; i < (t + 1) * E; i += NTD) {
    const int j = i / E, e = i - j * E;
    const float z = (qk_t[e] + ck[j]) * inv_sqrt_d;
    bp += fmaxf(z, 0.f); bn += fmaxf(-z, 0.f);
  }
  const float ibp = 1.0f / (block_sum(bp, red) + p.eps);
  const float ibn = 1.0f / (block_sum(bn, red) + p.eps);
  __syncthreads();
  // forward weights of THIS position → cache rows [t][n][i*E + e] (read by every later step of its descendants);
  // backward weights of this step → LDS
  float* wfa_t = p.wfa_c + ((long)t * NT + n) * TE;
  float* wfb_t = p.wfb_c + ((long)t * NT + n) * TE;
  for (int i = tid; i < (t + 1) * E; i += NTD) {
    const int j = i / E, e = i - j * E;
    const float q = j < t ? qkh[i] : qk_t[e];
    const float zf = (q + dk[j]) * inv_sqrt_d;
    const float zb = (qk_t[e] + ck[j]) * inv_sqrt_d;
    wfa_t[i] = fmaxf(zf, 0.f) * nfw[e];
    wfb_t[i] = fmaxf(-zf, 0.f) * nfw[MAX_E + e];
    wba[i] = fmaxf(zb, 0.f) * ibp;
    wbb[i] = fmaxf(-zb, 0.f) * ibn;
  }
  __syncthreads();                        // (also orders the block's own wfa_t / wfb_t stores before the reads below)
  // coefficients ca[i] = Σ_{j>=i} Σ_e wba[j][e]·wfa_j[i][e] (and cb): LP lanes per key position i (as many as the
  // block has for t+1 keys, 4..32), lane q takes j = i+q, i+q+LP, ...; fixed shuffle order → bit-identical run to run
  {
    int LP = 4;
    while (LP < 32 && (t + 1) * LP * 2 <= NTD) LP *= 2;
    const int i = tid / LP, q = tid - i * LP;
    if (i <= t) {
      float sa = 0.f, sb = 0.f;
      for (int j = i + q; j <= t; j += LP) {
        const long row = ((long)j * NT + slot[j]) * TE + (long)i * E;
        const float* fa = p.wfa_c + row;
        const float* fb = p.wfb_c + row;
        const float* ba = wba + j * E;
        const float* bb = wbb + j * E;
        for (int e = 0; e < E; e += 4) {
          const float4 x = *(const float4*)(fa + e), y = *(const float4*)(fb + e);
          sa = fmaf(ba[e], x.x, sa); sa = fmaf(ba[e + 1], x.y, sa); sa = fmaf(ba[e + 2], x.z, sa); sa = fmaf(ba[e + 3], x.w, sa);
          sb = fmaf(bb[e], y.x, sb); sb = fmaf(bb[e + 1], y.y, sb); sb = fmaf(bb[e + 2], y.z, sb); sb = fmaf(bb[e + 3], y.w, sb);
        }
      }
      for (int o = 1; o < LP; o <<= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); }
      if (q == 0) { scr[i] = sa; scr[T + i] = sb; }
    }
  }
  for (int j = tid; j <= t; j += NTD) {           // wja[j] = Σ_e wba[j][e]
    float sa = 0.f, sb = 0.f;
    for (int e = 0; e < E; ++e) { sa += wba[j * E + e]; sb += wbb[j * E + e]; }
    scr[2 * T + j] = sa; scr[3 * T + j] = sb;
  }
  {                                               // wea[e] = Σ_j wba[j][e]: 32 lanes per e
    const int e = tid >> 5, q = tid & 31;
    if (e < E) {
      float sa = 0.f, sb = 0.f;
      for (int j = q; j <= t; j += 32) { sa += wba[j * E + e]; sb += wbb[j * E + e]; }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); }
      if (q == 0) { scr[4 * T + e] = sa; scr[4 * T + E + e] = sb; }
    }
  }
  __syncthreads();

  // ---- phase 2: thread (half h, channel c): half 0 sums the earlier positions and the first E/2 bias rows, half 1
  //      the later ones; half 1's partial sums pass through LDS
  const float* ca = scr; const float* cb = scr + T;
  const float* wja = scr + 2 * T; const float* wjb = scr + 3 * T;
  const float* wea = scr + 4 * T; const float* web = wea + E;
  const bool valid = p.row_valid[n] != 0;
  constexpr int HALF = NTD / 2;
  const int h = tid / HALF, cl = tid - h * HALF;
  const int jm = (t + 1) / 2;
  const int j0 = h ? jm : 0, j1 = h ? t : jm;       // positions [j0, j1) from the caches; half 1 also takes position t
  const int e0 = h ? E / 2 : 0, e1 = h ? E : E / 2;
  for (int cb0 = 0; cb0 < d; cb0 += HALF) {
    const int c = cb0 + cl;
    float oa = 0.f, ob = 0.f;
    if (c < d) {
#pragma unroll 8
      for (int j = j0; j < j1; ++j) {
        const long o = ((long)j * NT + slot[j]) * d + c;
        const float va = p.va_c[o], vb = p.vb_c[o], cj = p.cond_c[o];
        oa = fmaf(ca[j], va, oa); oa = fmaf(wja[j], cj, oa);
        ob = fmaf(cb[j], vb, ob); ob = fmaf(wjb[j], cj, ob);
      }
      if (h) {
        oa = fmaf(ca[t], lin[2 * d + c], oa); oa = fmaf(wja[t], cond_t[c], oa);
        ob = fmaf(cb[t], lin[3 * d + c], ob); ob = fmaf(wjb[t], cond_t[c], ob);
      }
#pragma unroll 4
      for (int e = e0; e < e1; ++e) {
        const float be = p.bexp[(long)e * d + c];
        oa = fmaf(wea[e], be, oa);
        ob = fmaf(web[e], be, ob);
      }
    }
    if (h) { comb[cl] = oa; comb[HALF + cl] = ob; }
    __syncthreads();
    if (!h && c < d) {
      oa += comb[cl]; ob += comb[HALF + cl];
      float yv = p.y_in[(long)n * p.ldyi + c];
      if (valid) {
        const float sg = 1.0f / (1.0f + expf(-lin[4 * d + c]));
        yv += sg * oa + (1.0f - sg) * ob;
      }
      p.y[(long)n * p.ldy + c] = yv;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Cross attention: one block (4 waves) per (image, head, chunk of NB beams).  The k beams of an image
// attend to the SAME K/V, so every K/V element is loaded once per block and used for all NB queries
// (the one-wave-per-sequence form re-read 72 KB per beam and ran ~27 dependent load rounds):
//   scores : dk/16 lanes per key (16 floats each); the 4 waves sweep 4·64/(dk/16) keys at a time, all
//            sweeps of a wave in flight together; xor-shuffle reduce; NB dot products per loaded key
//   softmax: wave b normalises beam b (lanes stride over S), base e as the reference
//   P·V    : thread = (channel of the head, key group); coalesced dk·4-byte rows, 12 keys in flight,
//            NB accumulators; key groups combined through LDS.
// kv: [n_img, S, ldkv], K at koff, V at voff.
// ---------------------------------------------------------------------------------------------
// PF (diagnostic builds only, -DODIC_XATTN_VARIANTS, tools/xattn_ab.py): the software-pipelined P·V loop of round 2 —
// the first batch of V rows requested before the score phase, the next batch under the current batch's FMAs — whose
// results differed from this kernel's 30 times in 96,000 captions beside the encode graph.
//   PF = 1 as round 2 wrote it; 2..4 = the same with ONE change each, to locate the mechanism on the hardware:
//   2: `s_nop 7` after every group of P·V FMAs; 3: the FMAs kept scalar (no v_pk_fma_f32); 4: all of a key's
//   probabilities read and waited for (lgkmcnt(0)) before the first FMA.
template <int NB, int PF = 0>
__global__ __launch_bounds__(256) void cross_attn_step_kernel(const float* __restrict__ q, long ldq,
                                                              const float* __restrict__ kv, long ldkv, int koff,
                                                              int voff, const int* __restrict__ enc_len,
                                                              const int* __restrict__ row_valid,
                                                              float* __restrict__ out, long ldo, int beams, int S,
                                                              int d, int heads) {
  extern __shared__ float smem[];        // sc[NB][S] | red[256 / dk][NB][dk] | lsum[NB]
  const int img = blockIdx.x, h = blockIdx.y, b0 = blockIdx.z * NB;
  const int nb = min(NB, beams - b0);    // beams handled here
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int dk = d / heads;
  const int lpk = dk >> 4;               // lanes per key (1, 2 or 4)
  const int kps = 64 / lpk;              // keys per wave-sweep
  const int len = enc_len[img];
  const float inv = rsqrtf((float)dk);
  const float* kvb = kv + (long)img * S * ldkv;
  const int part = lane % lpk, kslot = lane / lpk;
  float* sc = smem;
  float* red = smem + NB * S;
  float* lsum = red + 256 * NB;
  const long n0 = (long)img * beams + b0;   // first sequence row of this block

  float4 qv[NB][4];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const float* qp = q + (n0 + min(b, nb - 1)) * ldq + h * dk + part * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) qv[b][i] = *(const float4*)(qp + 4 * i);
  }
  int valid[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) valid[b] = row_valid[n0 + min(b, nb - 1)];

  float vpre[12];
  if constexpr (PF) {       // V rows of the first P·V batch requested now: their latency passes under scores + softmax
    const int c_ = tid % dk, g_ = tid / dk, ng_ = 256 / dk;
    const float* vp_ = kvb + voff + h * dk + c_;
#pragma unroll
    for (int i = 0; i < 12; ++i) vpre[i] = vp_[(long)min(g_ + i * ng_, S - 1) * ldkv];
  }
  // ---- scores: wave w takes sweeps w, w+4, ...; up to 3 sweeps of K loads in flight
  const int nsweep = (S + kps - 1) / kps;
  for (int sw0 = wave; sw0 < nsweep; sw0 += 12) {
    float4 k4[3][4];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int s = min((sw0 + 4 * u) * kps + kslot, S - 1);
      const float* kr = kvb + (long)s * ldkv + koff + h * dk + part * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) k4[u][i] = *(const float4*)(kr + 4 * i);
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int sw = sw0 + 4 * u;
      const int s = sw * kps + kslot;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc = fmaf(qv[b][i].x, k4[u][i].x, acc); acc = fmaf(qv[b][i].y, k4[u][i].y, acc);
          acc = fmaf(qv[b][i].z, k4[u][i].z, acc); acc = fmaf(qv[b][i].w, k4[u][i].w, acc);
        }
        for (int o = 1; o < lpk; o <<= 1) acc += __shfl_xor(acc, o, 64);
        if (part == 0 && sw < nsweep && s < S) {
          float v = acc * inv;
          if (!valid[b] || s >= len) v = -1e4f;      // masked_fill(mask == 0, -1e4), layers.py:286
          sc[b * S + s] = v;
        }
      }
    }
  }
  __syncthreads();
  // ---- softmax: wave b → beam b, b + 4, ...
  for (int b = wave; b < nb; b += 4) {
    float m = -INFINITY;
    for (int s = lane; s < S; s += 64) m = fmaxf(m, sc[b * S + s]);
    m = wave_max(m);
    float l = 0.f;
    for (int s = lane; s < S; s += 64) { const float e = expf(sc[b * S + s] - m); sc[b * S + s] = e; l += e; }
    l = wave_sum(l);
    if (lane == 0) lsum[b] = l;
  }
  __syncthreads();
  // ---- P·V: thread = (channel c, key group g); groups take keys g, g + ng, ...
  const int c = tid % dk, g = tid / dk, ng = 256 / dk;
  float acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) acc[b] = 0.f;
  const float* vp = kvb + voff + h * dk + c;
  for (int s0 = g; s0 < S; s0 += 12 * ng) {
    float v[12];
    if constexpr (PF) {
#pragma unroll
      for (int i = 0; i < 12; ++i) v[i] = vpre[i];
      if (s0 + 12 * ng < S) {                        // next batch in flight under this one's FMAs
#pragma unroll
        for (int i = 0; i < 12; ++i) vpre[i] = vp[(long)min(s0 + 12 * ng + i * ng, S - 1) * ldkv];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 12; ++i) v[i] = vp[(long)min(s0 + i * ng, S - 1) * ldkv];
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int s = s0 + i * ng;
      if (s < S) {
        if constexpr (PF == 4) {
          float pb[NB];
#pragma unroll
          for (int b = 0; b < NB; ++b) pb[b] = sc[(b < nb ? b : 0) * S + s];
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pb[0]) :: "memory");
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[b] = fmaf(pb[b], v[i], acc[b]);
        } else {
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            acc[b] = fmaf(sc[(b < nb ? b : 0) * S + s], v[i], acc[b]);
            if constexpr (PF == 3) asm volatile("" : "+v"(acc[b]));
          }
          if constexpr (PF == 2) asm volatile("s_nop 7" ::: "memory");
        }
      }
    }
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) red[(g * NB + b) * dk + c] = acc[b];
  __syncthreads();
  for (int e = tid; e < nb * dk; e += 256) {
    const int b = e / dk, cc = e - b * dk;
    float v = 0.f;
    for (int gg = 0; gg < ng; ++gg) v += red[(gg * NB + b) * dk + cc];
    out[(n0 + b) * ldo + h * dk + cc] = v / lsum[b];
  }
}

// ---------------------------------------------------------------------------------------------
// log-softmax + exact top-k (ties → lower index) of up to NR rows AT ONCE by one 512-thread block (the fused search
// step: the k logits rows of one image; the stand-alone kernel: NR = 1).  A thread holds a 20-value slice of every
// row in registers (V <= 10240; longer rows stream).  Selection by threshold instead of sorting:
//   1. wave maxima of every row → LDS (their maximum is the row maximum the log-sum-exp needs anyway);
//   2. tau = the k-th largest of the 8 wave maxima: at least k elements are >= tau, so the k best all are;
//   3. in the Σexp pass every element >= tau (k .. a dozen of them) is appended to a candidate list in LDS;
//   4. wave r picks the k best of row r's candidates: k rounds of a wave arg-max over (value, lower index).
// Per row that is ~60 VALU operations per thread and k+1 shuffle chains, against ~1000 for per-thread sorted lists
// (which made a block that owns several rows ALU-bound).  Rows whose candidate list would overflow (k > 8 wave
// maxima, or hundreds of equal values) take k rounds of a block-wide arg-max instead — exact, just slower.
// `top_val` / `top_idx` ([row][k]) may point to global memory or LDS; logp0 (optional): the full log-prob rows.
// ---------------------------------------------------------------------------------------------
constexpr int TOPK_CAP = 128;          // candidates per row (two per lane of the selecting wave)
template <int NR> struct TopkSharedN {
  float red[NR][16]; float red2[NR][16]; int cnt[NR]; float cv[NR][TOPK_CAP]; int ci[NR][TOPK_CAP];
  float bv[16]; int bi[16];
};

__device__ __forceinline__ void wave_argmax(float& best, int& besti) {        // every lane ends with the winner
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
  }
}

template <int NR, int NTH, bool NORM>
__device__ __forceinline__ void rows_logsoftmax_topk(const float* __restrict__ x0, long ldl, int nr,
                                                     float* __restrict__ logp0, long ldp, float* top_val,
                                                     int* top_idx, int V, int k, TopkSharedN<NR>& sh) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NPT = 10240 / NTH, NWV = NTH / 64;
  const bool small = V <= NPT * NTH;
  float xv[NR][NPT];
  float tm[NR];
  if (small) {                                       // every row's loads are in flight before the first use
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const float* x = x0 + (long)min(r, nr - 1) * ldl;
#pragma unroll
      for (int u = 0; u < NPT; ++u) { const int i = tid + u * NTH; xv[r][u] = i < V ? x[i] : -INFINITY; }
    }
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    tm[r] = -INFINITY;
    if (r < nr) {
      if (small) {
#pragma unroll
        for (int u = 0; u < NPT; ++u) tm[r] = fmaxf(tm[r], xv[r][u]);
      } else {
        const float* x = x0 + (long)r * ldl;
        for (int i = tid; i < V; i += NTH) tm[r] = fmaxf(tm[r], x[i]);
      }
    }
  }
  __syncthreads();                                   // (the previous call's readers of sh are done)
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const float m = wave_max(tm[r]);
    if (lane == 0) sh.red[r][wave] = m;
  }
  if (tid < NR) sh.cnt[tid] = 0;
  __syncthreads();
  float mx[NR], tau[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    float wm[NWV];
#pragma unroll
    for (int i = 0; i < NWV; ++i) wm[i] = sh.red[r][i];
    float t = wm[0];
#pragma unroll
    for (int i = 1; i < NWV; ++i) t = fmaxf(t, wm[i]);
    mx[r] = t;
    tau[r] = -INFINITY;                              // k > NWV: every element is a candidate → the overflow path
    if (k <= NWV) {
#pragma unroll
      for (int i = 0; i < NWV; ++i) {                // the element with exactly k-1 others ranked above it
        int above = 0;
#pragma unroll
        for (int j = 0; j < NWV; ++j) above += (wm[j] > wm[i] || (wm[j] == wm[i] && j < i)) ? 1 : 0;
        if (above == k - 1) tau[r] = wm[i];
      }
    }
  }
  // Σ exp(x - max) and the candidates
  float sm[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    float sacc = 0.f;
    if (r < nr) {
      const float* x = x0 + (long)r * ldl;
      if (small) {
#pragma unroll
        for (int u = 0; u < NPT; ++u) {
          const float v = xv[r][u];
          if constexpr (NORM) sacc += expf(v - mx[r]);
          if (v >= tau[r] && tid + u * NTH < V) {
            const int pos = atomicAdd(&sh.cnt[r], 1);
            if (pos < TOPK_CAP) { sh.cv[r][pos] = v; sh.ci[r][pos] = tid + u * NTH; }
          }
        }
      } else {
        for (int i = tid; i < V; i += NTH) {
          const float v = x[i];
          if constexpr (NORM) sacc += expf(v - mx[r]);
          if (v >= tau[r]) {
            const int pos = atomicAdd(&sh.cnt[r], 1);
            if (pos < TOPK_CAP) { sh.cv[r][pos] = v; sh.ci[r][pos] = i; }
          }
        }
      }
    }
    sm[r] = wave_sum(sacc);
  }
#pragma unroll
  for (int r = 0; r < NR; ++r)
    if (lane == 0) sh.red2[r][wave] = sm[r];
  __syncthreads();
  float lse[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    lse[r] = 0.f;                                    // NORM = false: rows are log-probs already
    if constexpr (NORM) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < NWV; ++i) t += sh.red2[r][i];
      lse[r] = mx[r] + logf(t);
    }
  }
  if (logp0) {
#pragma unroll
    for (int r = 0; r < NR; ++r)
      if (r < nr)
        for (int i = tid; i < V; i += NTH) logp0[(long)r * ldp + i] = x0[(long)r * ldl + i] - lse[r];
  }
  // selection: wave r takes row r
  if (wave < nr) {
    const int r = wave, n = sh.cnt[r];
    if (n <= TOPK_CAP) {
      float l = lse[0];
#pragma unroll
      for (int q = 1; q < NR; ++q) if (r == q) l = lse[q];
      float c0 = -INFINITY, c1 = -INFINITY; int i0 = 0x7fffffff, i1 = 0x7fffffff;
      if (lane < n) { c0 = sh.cv[r][lane]; i0 = sh.ci[r][lane]; }
      if (lane + 64 < n) { c1 = sh.cv[r][lane + 64]; i1 = sh.ci[r][lane + 64]; }
      for (int rd = 0; rd < k; ++rd) {
        const bool first = c0 > c1 || (c0 == c1 && i0 < i1);
        float best = first ? c0 : c1; int besti = first ? i0 : i1;
        wave_argmax(best, besti);
        if (lane == 0) { top_val[r * k + rd] = best - l; top_idx[r * k + rd] = besti; }
        if (i0 == besti) { c0 = -INFINITY; i0 = 0x7fffffff; }
        if (i1 == besti) { c1 = -INFINITY; i1 = 0x7fffffff; }
      }
    }
  }
  // overflowed rows (block-uniform): k rounds of a block-wide arg-max over the elements ranked below the last winner
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    if (r < nr && sh.cnt[r] > TOPK_CAP) {
      const float* x = x0 + (long)r * ldl;
      float pv = INFINITY; int pi = -1;              // last winner: later rounds take (value, index) ranked after it
      for (int rd = 0; rd < k; ++rd) {
        float best = -INFINITY; int besti = 0x7fffffff;
        for (int i = tid; i < V; i += NTH) {
          const float v = x[i];
          const bool after = v < pv || (v == pv && i > pi);
          if (after && (v > best || (v == best && i < besti))) { best = v; besti = i; }
        }
        wave_argmax(best, besti);
        __syncthreads();
        if (lane == 0) { sh.bv[wave] = best; sh.bi[wave] = besti; }
        __syncthreads();
        best = sh.bv[0]; besti = sh.bi[0];
        for (int w = 1; w < NWV; ++w)
          if (sh.bv[w] > best || (sh.bv[w] == best && sh.bi[w] < besti)) { best = sh.bv[w]; besti = sh.bi[w]; }
        if (tid == 0) { top_val[r * k + rd] = best - lse[r]; top_idx[r * k + rd] = besti; }
        pv = best; pi = besti;
      }
    }
  }
  __syncthreads();
}

template <bool NORM = true>
__global__ __launch_bounds__(512) void logsoftmax_topk_kernel(const float* __restrict__ logits, long ldl,
                                                              float* __restrict__ logp_out, long ldp,
                                                              float* __restrict__ top_val, int* __restrict__ top_idx,
                                                              int V, int k) {
  __shared__ TopkSharedN<1> sh;
  const int n = blockIdx.x;
  rows_logsoftmax_topk<1, 512, NORM>(logits + (long)n * ldl, ldl, 1, logp_out ? logp_out + (long)n * ldp : nullptr, ldp,
                                     top_val + (long)n * k, top_idx + (long)n * k, V, k, sh);
}


// ---------------------------------------------------------------------------------------------
// Sampling without replacement from softmax(x) — the `sample` branches of the reference search
// (captioning_model.py:128-131,166-168: exp(log_probs).multinomial(k, replacement=False)) and the ancestral
// sampling of :59-109 (k = 1) — as ONE pass over the row: the k largest of  x[v] + Gumbel(v)  are distributed
// exactly as k sequential draws without replacement (Gumbel-top-k / Plackett-Luce), so no renormalisation loop
// and no host round trip.  Noise: Philox4x32-10 keyed by `seed`, counter = (row, vocabulary index / 4, *pos, 0)
// → reproducible for a given seed whatever the launch geometry; the position read from device memory makes a
// captured step graph draw fresh numbers at every replay.  Outputs: the chosen words in draw order (descending
// perturbed score) and their log-probabilities x[v] − logsumexp(x); optionally the full log-prob row.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float gumbel_from_bits(unsigned b) {
  const float u = ((float)(b >> 8) + 0.5f) * (1.0f / 16777216.0f);       // (0,1), 24 bits
  return -logf(-logf(u));
}

template <int KM>
__global__ __launch_bounds__(1024) void logsoftmax_sample_kernel(const float* __restrict__ logits, long ldl,
                                                                 float* __restrict__ logp_out, long ldp,
                                                                 float* __restrict__ top_val, int* __restrict__ top_idx,
                                                                 int V, int k, unsigned long long seed,
                                                                 const int* __restrict__ pos) {
  __shared__ float red[16];
  __shared__ float bv[16];
  __shared__ int bi[16];
  __shared__ int winner;
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* x = logits + (long)n * ldl;
  const unsigned step = pos ? (unsigned)*pos : 0u;
  float tv[KM]; int ti[KM];
#pragma unroll
  for (int q = 0; q < KM; ++q) { tv[q] = -INFINITY; ti[q] = 0x7fffffff; }
  float mx = -INFINITY;
  // a thread owns groups of 4 consecutive words: one Philox call per group
  for (int g4 = tid; g4 * 4 < V; g4 += 1024) {
    unsigned r[4];
    philox4x32_10((unsigned)n, (unsigned)g4, step, 0u, (unsigned)seed, (unsigned)(seed >> 32), r);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int vi0 = g4 * 4 + e;
      if (vi0 < V) {
        const float xv = x[vi0];
        mx = fmaxf(mx, xv);
        float v = xv + gumbel_from_bits(r[e]); int vi = vi0;
        if (v > tv[KM - 1]) {
#pragma unroll
          for (int q = 0; q < KM; ++q) {
            if (v > tv[q]) { const float fv = tv[q]; const int fi = ti[q]; tv[q] = v; ti[q] = vi; v = fv; vi = fi; }
          }
        }
      }
    }
  }
  const float m = block_max(mx, red);
  float s = 0.f;
  for (int i = tid; i < V; i += 1024) s += expf(x[i] - m);
  s = block_sum(s, red);
  const float lse = m + logf(s);
  if (logp_out)
    for (int i = tid; i < V; i += 1024) logp_out[(long)n * ldp + i] = x[i] - lse;
  for (int r = 0; r < k; ++r) {
    float best = tv[0]; int besti = ti[0];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(besti, o, 64);
      if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if (lane == 0) { bv[wave] = best; bi[wave] = besti; }
    __syncthreads();
    if (tid == 0) {
      float b = bv[0]; int ix = bi[0];
      for (int w = 1; w < 16; ++w)
        if (bv[w] > b || (bv[w] == b && bi[w] < ix)) { b = bv[w]; ix = bi[w]; }
      winner = ix;
      top_val[(long)n * k + r] = x[ix] - lse;            // the word's log-probability, not its perturbed score
      top_idx[(long)n * k + r] = ix;
    }
    __syncthreads();
    if (ti[0] == winner) {
#pragma unroll
      for (int q = 0; q + 1 < KM; ++q) { tv[q] = tv[q + 1]; ti[q] = ti[q + 1]; }
      tv[KM - 1] = -INFINITY; ti[KM - 1] = 0x7fffffff;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Beam bookkeeping: one block per image (captioning_model.py:172-223).  Wave 0 does the k·k selection — one lane per
// candidate (four at k = 16), k rounds of a wave arg-max in which the lowest flat index wins a tie (the scan order
// of the reference's topk over the flattened k x k scores) — then all threads permute the prefix / log-prob /
// ancestor rows IN PLACE (a thread owns whole columns j: it reads the k parent values of its column first, then
// writes the k new rows) and, when asked to, write the embedding of the chosen words for the next position
// (EmbeddingLayer, layers.py:118-121: what odic_dec_embed would do in its own launch).  The block that arrives last
// at `ctr` (low 16 bits = arrivals, high bits = images with a still-growing beam) advances *pos, raises *done when
// nothing grows any more, and re-arms the counter.
// ---------------------------------------------------------------------------------------------
struct BeamParams {
  const float* cand_val; const int* cand_idx;
  long long* tok; float* lp; int* anc;
  float* cumul; int* n_elem; int* has_eos; int* row_valid; long long* next_tok;
  int* pos; int* done; int* ctr;
  int n_img, k, T; long long eos;
};
struct EmbedArgs { const float* embed; const float* pos_table; float* y; long ldy; int d; float scale; int pos_rows; };

struct BeamShared {
  float cv[MAX_K * MAX_K]; int ci[MAX_K * MAX_K];      // candidates: log-prob, word  [beam][rank]
  int eos[MAX_K]; int ne[MAX_K]; float cu[MAX_K];
  float lpm[MAX_K][MAX_T];                             // per-token log-probs of the k prefixes
  int parent[MAX_K]; int word[MAX_K]; float lp[MAX_K]; float cumul[MAX_K];
};

// s.cv / s.ci hold the k x k candidates (written by this block; a barrier follows inside)
template <int NT>
__device__ __forceinline__ void beam_update(const BeamParams& p, const EmbedArgs& e, BeamShared& s, int b, int t) {
  const int tid = threadIdx.x, k = p.k, T = p.T;
  for (int r = tid; r < k; r += NT) {
    s.eos[r] = p.has_eos[b * k + r]; s.ne[r] = p.n_elem[b * k + r]; s.cu[r] = p.cumul[b * k + r];
  }
  for (int i = tid; i < k * (t + 1); i += NT) {
    const int r = i / (t + 1), j = i - r * (t + 1);
    s.lpm[r][j] = p.lp[((long)b * k + r) * T + j];
  }
  __syncthreads();

  if (tid < 64) {
    const int lane = tid;
    if (t == 0) {                       // seeding: the k best words of beam 0
      if (lane < k) { s.parent[lane] = 0; s.word[lane] = s.ci[lane]; s.lp[lane] = s.cv[lane]; }
    } else {
      constexpr int CPL = MAX_K * MAX_K / 64;        // candidates per lane
      float tot[CPL], val[CPL];
#pragma unroll
      for (int u = 0; u < CPL; ++u) {
        const int i = lane + 64 * u;
        tot[u] = -INFINITY; val[u] = 0.f;
        if (i < k * k) {
          const int j = i / k, c = i - j * k;
          val[u] = s.eos[j] ? (c == 0 ? 0.0f : -999.0f) : s.cv[i];
          tot[u] = s.cu[j] + val[u];
        }
      }
      for (int r = 0; r < k; ++r) {
        float best = tot[0], bval = val[0]; int bi = lane;
#pragma unroll
        for (int u = 1; u < CPL; ++u)
          if (tot[u] > best) { best = tot[u]; bval = val[u]; bi = lane + 64 * u; }
        if (!(best > -INFINITY)) bi = 0x7fffffff;               // nothing left in this lane
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const float ov = __shfl_xor(best, o, 64), ol = __shfl_xor(bval, o, 64);
          const int oi = __shfl_xor(bi, o, 64);
          if (ov > best || (ov == best && oi < bi)) { best = ov; bval = ol; bi = oi; }
        }
#pragma unroll
        for (int u = 0; u < CPL; ++u)
          if (bi == lane + 64 * u) tot[u] = -INFINITY;
        if (lane == 0) { s.parent[r] = bi / k; s.word[r] = s.ci[bi]; s.lp[r] = bval; }
      }
    }
  }
  __syncthreads();
  // cumulative score = re-summed per-token log-probs of the parent prefix + the new one (:213), in position order
  if (tid < k) {
    const float* src = s.lpm[s.parent[tid]];
    float cs = 0.f;
    for (int j = 0; j <= t; ++j) cs += src[j];
    s.cumul[tid] = cs + s.lp[tid];
  }
  const long base = (long)b * k * T;
  if constexpr (NT >= 256) {
    // one (row, column) element per thread and pass: all reads of the old rows, a barrier, then the writes
    constexpr int EPT = (MAX_K * MAX_T + NT - 1) / NT;
    long long tv[EPT]; float lv[EPT]; int av[EPT];
    const int cols = t + 1;
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
      const int i = tid + u * NT;
      if (i < k * cols) {
        const int r = i / cols, j = i - r * cols;
        const long src = base + (long)s.parent[r] * T + j;
        tv[u] = p.tok[src]; lv[u] = p.lp[src];
        av[u] = j < t ? p.anc[src] : b * k + s.parent[r];
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
      const int i = tid + u * NT;
      if (i < k * cols) {
        const int r = i / cols, j = i - r * cols;
        const long dst = base + (long)r * T + j;
        p.tok[dst] = tv[u]; p.lp[dst] = lv[u]; p.anc[dst] = av[u];
      }
    }
  } else {
    for (int j = tid; j <= t; j += NT) {                // a thread owns whole columns
      long long tv[MAX_K]; float lv[MAX_K]; int av[MAX_K];
#pragma unroll
      for (int r = 0; r < MAX_K; ++r) {
        if (r < k) {
          const long src = base + (long)s.parent[r] * T + j;
          tv[r] = p.tok[src]; lv[r] = p.lp[src];
          av[r] = j < t ? p.anc[src] : b * k + s.parent[r];
        }
      }
#pragma unroll
      for (int r = 0; r < MAX_K; ++r) {
        if (r < k) {
          const long dst = base + (long)r * T + j;
          p.tok[dst] = tv[r]; p.lp[dst] = lv[r]; p.anc[dst] = av[r];
        }
      }
    }
  }
  if (e.embed && t + 2 < T && t + 1 < e.pos_rows) {     // input of the next position for the k chosen words (the
                                                        // last prefix position T-1 is never fed back; never past the table)
    const float* prow = e.pos_table + (long)(t + 1) * e.d;
    for (int i = tid; i < k * e.d; i += NT) {
      const int r = i / e.d, c = i - r * e.d;
      e.y[(long)(b * k + r) * e.ldy + c] = e.embed[(long)s.word[r] * e.d + c] * e.scale + prow[c];
    }
  }
  __syncthreads();                                      // s.cumul; every thread's reads of the old rows are done
  int alive = 0;
  if (tid < k) {
    const int r = tid, par = s.parent[r];
    const int pe = t == 0 ? 0 : s.eos[par];
    const int ne = t == 0 ? 1 : s.ne[par];
    const long dst = base + (long)r * T;
    p.tok[dst + t + 1] = (long long)s.word[r];
    p.lp[dst + t + 1] = s.lp[r];
    p.cumul[b * k + r] = s.cumul[r];
    p.n_elem[b * k + r] = ne + (pe ? 0 : 1);
    p.has_eos[b * k + r] = (pe || ((long long)s.word[r] == p.eos)) ? 1 : 0;
    p.row_valid[b * k + r] = pe ? 0 : 1;
    p.next_tok[b * k + r] = (long long)s.word[r];
    alive = pe ? 0 : 1;
  }
  if (tid < 64) {
    const int alive_any = __any(alive) ? 1 : 0;         // (k <= 16 lanes of wave 0 carry the flags)
    if (tid == 0) {
      const int prev = atomicAdd(p.ctr, 1 + (alive_any << 16));
      if ((prev & 0xffff) == p.n_img - 1) {             // every block has read *pos before arriving here
        const int al = (prev >> 16) + alive_any;
        if (!al) *p.done = 1;
        *p.pos = t + 1;
        *p.ctr = 0;
      }
    }
  }
}

__global__ __launch_bounds__(256) void beam_step_kernel(BeamParams p, EmbedArgs e) {
  __shared__ BeamShared s;
  const int b = blockIdx.x, k = p.k;
  const int t = *p.pos;                 // position just processed; prefix length is t+1
  if (t + 1 >= p.T) return;             // the prefix is full: a replay past the last step changes nothing
  for (int i = threadIdx.x; i < k * k; i += 256) {
    s.cv[i] = p.cand_val[(long)b * k * k + i];
    s.ci[i] = p.cand_idx[(long)b * k * k + i];
  }
  beam_update<256>(p, e, s, b, t);       // (four waves: the k·d embedding rows and the prefix re-gather are the bulk)
}

// The whole tail of a search step in one launch: log-softmax + top-k of the image's k logits rows, NR rows per pass
// (the candidates stay in LDS), the beam update above and the next position's input embedding.
template <int NR>
__global__ __launch_bounds__(512) void beam_search_step_kernel(const float* __restrict__ logits, long ldl, int V,
                                                                BeamParams p, EmbedArgs e) {
  __shared__ BeamShared s;
  __shared__ TopkSharedN<NR> sh;
  const int b = blockIdx.x, k = p.k;
  const int t = *p.pos;
  if (t + 1 >= p.T) return;
  const int nrows = t == 0 ? 1 : k;     // at the first position only beam 0 seeds the search
  for (int r0 = 0; r0 < nrows; r0 += NR)
    rows_logsoftmax_topk<NR, 512, true>(logits + (long)(b * k + r0) * ldl, ldl, min(NR, nrows - r0), nullptr, 0, s.cv + r0 * k,
                                 s.ci + r0 * k, V, k, sh);
  beam_update<512>(p, e, s, b, t);
}

__global__ void beam_finalize_kernel(const float* cumul, const int* n_elem, int* order, float* score, int n_img,
                                     int k) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_img) return;
  float sc[MAX_K];
  for (int j = 0; j < k; ++j) { sc[j] = cumul[b * k + j] / (float)n_elem[b * k + j]; score[b * k + j] = sc[j]; }
  for (int r = 0; r < k; ++r) {
    int bi = 0; float bv = -INFINITY;
    for (int j = 0; j < k; ++j) if (sc[j] > bv) { bv = sc[j]; bi = j; }
    sc[bi] = -INFINITY;
    order[b * k + r] = bi;
  }
}

// odic_beam_finalize + emission of the best caption (EOS-padded int32 row + length): one 64-lane block per image
__global__ __launch_bounds__(64) void beam_finalize_best_kernel(const float* cumul, const int* n_elem,
                                                                const long long* tok, int* order, float* score,
                                                                int* out_tok, int* out_len, int k, int T, int pad) {
  __shared__ int s_best;
  const int b = blockIdx.x, lane = threadIdx.x;
  if (lane == 0) {
    float sc[MAX_K];
    for (int j = 0; j < k; ++j) { sc[j] = cumul[b * k + j] / (float)n_elem[b * k + j]; score[b * k + j] = sc[j]; }
    for (int r = 0; r < k; ++r) {
      int bi = 0; float bv = -INFINITY;
      for (int j = 0; j < k; ++j) if (sc[j] > bv) { bv = sc[j]; bi = j; }
      sc[bi] = -INFINITY;
      order[b * k + r] = bi;
      if (r == 0) s_best = bi;
    }
  }
  __syncthreads();
  const int best = s_best;
  const int len = n_elem[b * k + best];
  const long long* src = tok + ((long)b * k + best) * T;
  for (int j = lane; j < T; j += 64) out_tok[(long)b * T + j] = j < len ? (int)src[j] : pad;
  if (lane == 0) out_len[b] = len;
}

__global__ void beam_reset_kernel(long long* tok, float* lp, int* row_valid, long long* next_tok, int* pos,
                                  int* done, int* ctr, int N, int T, long long sos, EmbedArgs e) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) { tok[(long)i * T] = sos; lp[(long)i * T] = 0.f; row_valid[i] = 1; next_tok[i] = sos; }
  if (i == 0) { *pos = 0; *done = 0; *ctr = 0; }
  if (e.embed)                                          // input of position 0: the start token
    for (long q = i; q < (long)N * e.d; q += (long)gridDim.x * blockDim.x) {
      const long r = q / e.d; const int c = (int)(q - r * e.d);
      e.y[r * e.ldy + c] = e.embed[sos * e.d + c] * e.scale + e.pos_table[c];
    }
}

}  // namespace

extern "C" int odic_dec_embed(const int64_t* tokens, const float* embed, const float* pos_table,
                              const int32_t* pos, float* y, int64_t ldy, int32_t N, int32_t d, int32_t pos_rows,
                              float scale, void* stream) {
  if (!tokens || !embed || !pos_table || !pos || !y) return ODIC_ENULL;
  if (N <= 0 || d <= 0 || pos_rows <= 0) return ODIC_EINVAL;
  hipLaunchKernelGGL(dec_embed_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, (const long long*)tokens, embed,
                     pos_table, pos, y, (long)ldy, N, d, pos_rows, scale);
  return odic_launch_status();
}

extern "C" int odic_dynexp_step(const float* lin, int64_t ldlin, const float* qexp, const float* bexp,
                                float* cond_c, float* key_c, float* va_c, float* vb_c, float* wfa_c,
                                float* wfb_c, float* qk_c, const int32_t* anc, const int32_t* row_valid,
                                const int32_t* pos, const float* y_in, int64_t ldy_in, float* y, int64_t ldy,
                                int32_t N, int32_t T, int32_t d, int32_t E, float eps, void* stream) {
  if (!lin || !qexp || !bexp || !cond_c || !key_c || !va_c || !vb_c || !wfa_c || !wfb_c || !qk_c || !anc ||
      !row_valid || !pos || !y || !y_in)
    return ODIC_ENULL;
  if (N <= 0 || T <= 0 || T > MAX_T || d <= 0 || E <= 0 || E > MAX_E) return ODIC_EINVAL;
  if (E != 4 && E != 8 && E != 16 && E != 32) return ODIC_EUNSUPPORTED;
  if (d % 64) return ODIC_EINVAL;
  DynParams p;
  p.lin = lin; p.ldlin = ldlin; p.qexp = qexp; p.bexp = bexp; p.cond_c = cond_c; p.key_c = key_c; p.va_c = va_c;
  p.vb_c = vb_c; p.wfa_c = wfa_c; p.wfb_c = wfb_c; p.qk_c = qk_c; p.anc = anc; p.row_valid = row_valid;
  p.pos = pos; p.y_in = y_in; p.ldyi = ldy_in; p.y = y; p.ldy = ldy;
  p.N = N; p.T = T; p.d = d; p.E = E; p.eps = eps;
  const size_t shmem = (size_t)(2 * d + 2 * T + 3 * MAX_E + 16 + 3 * T * E + 4 * T + 2 * E + DYN_NT) * sizeof(float) +
                       (size_t)T * sizeof(int);
  if (shmem > 64 * 1024) return ODIC_EINVAL;
  hipLaunchKernelGGL(dynexp_step_kernel, dim3(N), dim3(DYN_NT), shmem, (hipStream_t)stream, p);
  return odic_launch_status();
}

extern "C" int odic_cross_attn_step(const float* q, int64_t ldq, const float* kv, int64_t ldkv, int32_t koff,
                                    int32_t voff, const int32_t* enc_len, const int32_t* row_valid, float* out,
                                    int64_t ldo, int32_t N, int32_t n_img, int32_t S, int32_t d, int32_t heads,
                                    void* stream) {
  if (!q || !kv || !enc_len || !row_valid || !out) return ODIC_ENULL;
  if (N <= 0 || n_img <= 0 || N % n_img || S <= 0 || d <= 0 || heads <= 0 || d % heads) return ODIC_EINVAL;
  const int dk = d / heads;
  if (dk != 16 && dk != 32 && dk != 64) return ODIC_EUNSUPPORTED;
  if ((ldq & 3) || (ldkv & 3) || (koff & 3) || ((uintptr_t)q & 15) || ((uintptr_t)kv & 15)) return ODIC_EINVAL;
  if (S > 2048) return ODIC_EINVAL;       // score rows of up to 5 beams live in LDS
  const int beams = N / n_img;
  hipStream_t st = (hipStream_t)stream;
#define ODIC_XATTN(NB)                                                                                              \
  hipLaunchKernelGGL(cross_attn_step_kernel<NB>, dim3(n_img, heads, (beams + NB - 1) / NB), dim3(256),              \
                     (size_t)(NB * S + 256 * NB + NB) * sizeof(float), st, q, (long)ldq, kv, (long)ldkv, koff, voff,  \
                     enc_len, row_valid, out, (long)ldo, beams, S, d, heads)
  if (heads > 65535) return ODIC_EINVAL;
  if (beams == 1) ODIC_XATTN(1);
  else if (beams == 2) ODIC_XATTN(2);
  else if (beams == 3) ODIC_XATTN(3);
  else if (beams == 5 || beams > 8) ODIC_XATTN(5);
  else ODIC_XATTN(4);
#undef ODIC_XATTN
  return odic_launch_status();
}

// Ensemble step distribution (ensemble_captioning_model.py:66-83): out[n][v] = log(mean_m softmax(logits_m[n])[v]).
// One block per row; per model the row max and Σexp, then one pass that averages the probabilities.
constexpr int MAX_MODELS = 8;
struct EnsembleParams { const float* logits[MAX_MODELS]; int M; long ldl; float* out; long ldo; int V; };

__global__ __launch_bounds__(1024) void ensemble_logprob_kernel(EnsembleParams p) {
  __shared__ float red[16];
  __shared__ float s_max[MAX_MODELS], s_inv[MAX_MODELS];
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int m = 0; m < p.M; ++m) {
    const float* x = p.logits[m] + (long)n * p.ldl;
    float mx = -INFINITY;
    for (int i = tid; i < p.V; i += 1024) mx = fmaxf(mx, x[i]);
    mx = block_max(mx, red);
    float s = 0.f;
    for (int i = tid; i < p.V; i += 1024) s += expf(x[i] - mx);
    s = block_sum(s, red);
    if (tid == 0) { s_max[m] = mx; s_inv[m] = 1.0f / s; }
    __syncthreads();
  }
  const float inv_m = 1.0f / (float)p.M;
  for (int i = tid; i < p.V; i += 1024) {
    float acc = 0.f;
    for (int m = 0; m < p.M; ++m) acc += expf(p.logits[m][(long)n * p.ldl + i] - s_max[m]) * s_inv[m];
    p.out[(long)n * p.ldo + i] = logf(acc * inv_m);
  }
}

extern "C" int odic_ensemble_logprobs(const float* const* logits, int32_t M, int64_t ldl, float* out, int64_t ldo,
                                      int32_t N, int32_t V, void* stream) {
  if (!logits || !out) return ODIC_ENULL;
  if (M <= 0 || M > MAX_MODELS || N <= 0 || V <= 0) return ODIC_EINVAL;
  EnsembleParams p;
  for (int m = 0; m < MAX_MODELS; ++m) p.logits[m] = m < M ? logits[m] : nullptr;
  for (int m = 0; m < M; ++m) if (!p.logits[m]) return ODIC_ENULL;
  p.M = M; p.ldl = ldl; p.out = out; p.ldo = ldo; p.V = V;
  hipLaunchKernelGGL(ensemble_logprob_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, p);
  return odic_launch_status();
}

extern "C" int odic_topk_rows(const float* logp, int64_t ldl, float* top_val, int32_t* top_idx, int32_t N, int32_t V,
                              int32_t k, void* stream) {
  if (!logp || !top_val || !top_idx) return ODIC_ENULL;
  if (N <= 0 || V <= 0 || k <= 0 || k > MAX_K || k > V) return ODIC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(logsoftmax_topk_kernel<false>, dim3(N), dim3(512), 0, s, logp, (long)ldl, (float*)nullptr, 0L, top_val, top_idx, V, k);
  return odic_launch_status();
}

extern "C" int odic_logsoftmax_topk(const float* logits, int64_t ldl, float* logp_out, int64_t ldp, float* top_val,
                                    int32_t* top_idx, int32_t N, int32_t V, int32_t k, void* stream) {
  if (!logits || !top_val || !top_idx) return ODIC_ENULL;
  if (N <= 0 || V <= 0 || k <= 0 || k > MAX_K || k > V) return ODIC_EINVAL;
  hipLaunchKernelGGL(logsoftmax_topk_kernel<true>, dim3(N), dim3(512), 0, (hipStream_t)stream, logits, (long)ldl, logp_out,
                     (long)ldp, top_val, top_idx, V, k);
  return odic_launch_status();
}

static int beam_params(BeamParams& p, EmbedArgs& e, const odic_beam_state* st, const odic_embed_args* emb,
                       int32_t n_img, int32_t beams, int32_t T, int64_t eos_idx) {
  if (!st) return ODIC_ENULL;
  // T: k x (t+1) per-token log-probs are staged in LDS [MAX_K][MAX_T]; n_img: `alive << 16` in the counter
  if (n_img <= 0 || beams <= 0 || beams > MAX_K || T <= 1 || T > MAX_T || n_img > 32767) return ODIC_EINVAL;
  if (!st->tokens || !st->logprobs || !st->anc || !st->cumul || !st->n_elem || !st->has_eos || !st->row_valid ||
      !st->next_tok || !st->pos || !st->done || !st->ctr)
    return ODIC_ENULL;
  p.cand_val = nullptr; p.cand_idx = nullptr;
  p.tok = (long long*)st->tokens; p.lp = st->logprobs; p.anc = st->anc;
  p.cumul = st->cumul; p.n_elem = st->n_elem; p.has_eos = st->has_eos; p.row_valid = st->row_valid;
  p.next_tok = (long long*)st->next_tok; p.pos = st->pos; p.done = st->done; p.ctr = st->ctr;
  p.n_img = n_img; p.k = beams; p.T = T; p.eos = eos_idx;
  e.embed = nullptr; e.pos_table = nullptr; e.y = nullptr; e.ldy = 0; e.d = 0; e.scale = 0.f; e.pos_rows = 0;
  if (emb) {
    if (!emb->embed || !emb->pos_table || !emb->y) return ODIC_ENULL;
    if (emb->d <= 0 || emb->pos_rows <= 0) return ODIC_EINVAL;
    e.embed = emb->embed; e.pos_table = emb->pos_table; e.y = emb->y; e.ldy = emb->ldy; e.d = emb->d; e.scale = emb->scale;
    e.pos_rows = emb->pos_rows;
  }
  return 0;
}

extern "C" int odic_beam_step(const float* cand_val, const int32_t* cand_idx, const odic_beam_state* st,
                              const odic_embed_args* emb, int32_t n_img, int32_t beams, int32_t T, int64_t eos_idx,
                              void* stream) {
  if (!cand_val || !cand_idx) return ODIC_ENULL;
  BeamParams p; EmbedArgs e;
  const int rc = beam_params(p, e, st, emb, n_img, beams, T, eos_idx);
  if (rc != 0) return rc;
  p.cand_val = cand_val; p.cand_idx = cand_idx;
  hipLaunchKernelGGL(beam_step_kernel, dim3(n_img), dim3(256), 0, (hipStream_t)stream, p, e);
  return odic_launch_status();
}

extern "C" int odic_beam_search_step(const float* logits, int64_t ldl, int32_t V, const odic_beam_state* st,
                                     const odic_embed_args* emb, int32_t n_img, int32_t beams, int32_t T,
                                     int64_t eos_idx, void* stream) {
  if (!logits) return ODIC_ENULL;
  BeamParams p; EmbedArgs e;
  const int rc = beam_params(p, e, st, emb, n_img, beams, T, eos_idx);
  if (rc != 0) return rc;
  if (V <= 0 || beams > V) return ODIC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  // rows per pass: a thread's 20-value slice of every row of the pass lives in registers
  if (beams <= 3) hipLaunchKernelGGL(beam_search_step_kernel<3>, dim3(n_img), dim3(512), 0, s, logits, (long)ldl, V, p, e);
  else if (beams <= 5) hipLaunchKernelGGL(beam_search_step_kernel<5>, dim3(n_img), dim3(512), 0, s, logits, (long)ldl, V, p, e);
  else hipLaunchKernelGGL(beam_search_step_kernel<6>, dim3(n_img), dim3(512), 0, s, logits, (long)ldl, V, p, e);
  return odic_launch_status();
}

extern "C" int odic_beam_finalize(const odic_beam_state* st, int32_t* order, float* score, int32_t n_img,
                                  int32_t beams, void* stream) {
  if (!st || !order || !score) return ODIC_ENULL;
  if (n_img <= 0 || beams <= 0 || beams > MAX_K) return ODIC_EINVAL;
  hipLaunchKernelGGL(beam_finalize_kernel, dim3((n_img + 63) / 64), dim3(64), 0, (hipStream_t)stream, st->cumul,
                     st->n_elem, order, score, n_img, beams);
  return odic_launch_status();
}

extern "C" int odic_beam_finalize_best(const odic_beam_state* st, int32_t* order, float* score, int32_t* out_tok,
                                       int32_t* out_len, int32_t n_img, int32_t beams, int32_t T, int32_t pad_idx,
                                       void* stream) {
  if (!st || !order || !score || !out_tok || !out_len) return ODIC_ENULL;
  if (!st->tokens || !st->cumul || !st->n_elem) return ODIC_ENULL;
  if (n_img <= 0 || beams <= 0 || beams > MAX_K || T <= 0) return ODIC_EINVAL;
  hipLaunchKernelGGL(beam_finalize_best_kernel, dim3(n_img), dim3(64), 0, (hipStream_t)stream, st->cumul, st->n_elem,
                     (const long long*)st->tokens, order, score, out_tok, out_len, beams, T, pad_idx);
  return odic_launch_status();
}

extern "C" int odic_beam_reset(const odic_beam_state* st, const odic_embed_args* emb, int32_t n_img, int32_t beams,
                               int32_t T, int64_t sos_idx, void* stream) {
  if (!st) return ODIC_ENULL;
  EmbedArgs e; e.embed = nullptr; e.pos_table = nullptr; e.y = nullptr; e.ldy = 0; e.d = 0; e.scale = 0.f; e.pos_rows = 0;
  if (emb) {
    if (!emb->embed || !emb->pos_table || !emb->y) return ODIC_ENULL;
    if (emb->d <= 0 || emb->pos_rows <= 0) return ODIC_EINVAL;
    e.embed = emb->embed; e.pos_table = emb->pos_table; e.y = emb->y; e.ldy = emb->ldy; e.d = emb->d; e.scale = emb->scale;
    e.pos_rows = emb->pos_rows;
  }
  if (!st->tokens || !st->logprobs || !st->row_valid || !st->next_tok || !st->pos || !st->done || !st->ctr)
    return ODIC_ENULL;
  if (n_img <= 0 || beams <= 0 || beams > MAX_K || T <= 0) return ODIC_EINVAL;
  const int N = n_img * beams;
  const int nblk = emb ? (int)(((long)N * e.d + 255) / 256 < 1024 ? ((long)N * e.d + 255) / 256 : 1024) : (N + 255) / 256;
  hipLaunchKernelGGL(beam_reset_kernel, dim3(nblk < (N + 255) / 256 ? (N + 255) / 256 : nblk), dim3(256), 0,
                     (hipStream_t)stream, (long long*)st->tokens, st->logprobs, st->row_valid,
                     (long long*)st->next_tok, st->pos, st->done, st->ctr, N, T, (long long)sos_idx, e);
  return odic_launch_status();
}

extern "C" int odic_logsoftmax_sample(const float* logits, int64_t ldl, float* logp_out, int64_t ldp, float* top_val,
                                      int32_t* top_idx, int32_t N, int32_t V, int32_t k, uint64_t seed,
                                      const int32_t* pos, void* stream) {
  if (!logits || !top_val || !top_idx) return ODIC_ENULL;
  if (N <= 0 || V <= 0 || k <= 0 || k > MAX_K || k > V) return ODIC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (k <= 4) hipLaunchKernelGGL(logsoftmax_sample_kernel<4>, dim3(N), dim3(1024), 0, s, logits, (long)ldl, logp_out, (long)ldp, top_val, top_idx, V, k, (unsigned long long)seed, pos);
  else if (k <= 8) hipLaunchKernelGGL(logsoftmax_sample_kernel<8>, dim3(N), dim3(1024), 0, s, logits, (long)ldl, logp_out, (long)ldp, top_val, top_idx, V, k, (unsigned long long)seed, pos);
  else hipLaunchKernelGGL(logsoftmax_sample_kernel<16>, dim3(N), dim3(1024), 0, s, logits, (long)ldl, logp_out, (long)ldp, top_val, top_idx, V, k, (unsigned long long)seed, pos);
  return odic_launch_status();
}

#ifdef ODIC_XATTN_VARIANTS
// =================================================================================================
// Diagnostic build only (tools/xattn_ab.py → tools/_build/libodic_dbg.so; never part of libodic_hip.so): the two forms
// of the cross-attention step side by side on the SAME inputs inside the real pipeline, every output element compared
// on the device, and the operands of the first mismatch kept for the post-mortem.
// =================================================================================================
namespace {
struct DbgState { int count; int first_site; int row; int col; int snapped; int launches; int per_variant[10]; };

__global__ __launch_bounds__(256) void dbg_compare_kernel(const float* __restrict__ a, const float* __restrict__ b, int n,
                                                          int ld, int d, int site, DbgState* st, int variant) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0 && variant == 1) atomicAdd(&st->launches, 1);
  if (i >= n) return;
  const int r = i / d, c = i - r * d;
  const unsigned x = __float_as_uint(a[(long)r * ld + c]), y = __float_as_uint(b[(long)r * ld + c]);
  if (x != y) {
    atomicAdd(&st->per_variant[variant], 1);
    if (variant == 1) {
      atomicAdd(&st->count, 1);
      if (atomicCAS(&st->first_site, -1, site) == -1) { st->row = r; st->col = c; }
    }
  }
}
// one block: if this site holds the first mismatch and nothing was kept yet, keep q / both outputs of the image's beams
// and the K / V columns [koff, koff + d) / [voff, voff + d) of the image
__global__ __launch_bounds__(256) void dbg_snapshot_kernel(const float* q, long ldq, const float* kv, long ldkv, int koff,
                                                           int voff, const float* a, const float* b, long ldo, int beams,
                                                           int S, int d, int site, DbgState* st, float* snap) {
  if (st->first_site != site || st->snapped) return;
  __syncthreads();
  const int img = st->row / beams;
  float* sq = snap; float* sa = sq + beams * d; float* sb = sa + beams * d; float* sk = sb + beams * d; float* sv = sk + (long)S * d;
  for (int i = threadIdx.x; i < beams * d; i += 256) {
    const int r = i / d, c = i - r * d;
    sq[i] = q[(long)(img * beams + r) * ldq + c];
    sa[i] = a[(long)(img * beams + r) * ldo + c];
    sb[i] = b[(long)(img * beams + r) * ldo + c];
  }
  for (int i = threadIdx.x; i < S * d; i += 256) {
    const int s_ = i / d, c = i - s_ * d;
    sk[i] = kv[((long)img * S + s_) * ldkv + koff + c];
    sv[i] = kv[((long)img * S + s_) * ldkv + voff + c];
  }
  __syncthreads();
  if (threadIdx.x == 0) st->snapped = 1;
}
}  // namespace

// variant 0 = the shipped kernel, 1 = the round-2 software-pipelined form
extern "C" int odic_dbg_cross_attn_step(int variant, const float* q, int64_t ldq, const float* kv, int64_t ldkv, int32_t koff,
                                        int32_t voff, const int32_t* enc_len, const int32_t* row_valid, float* out,
                                        int64_t ldo, int32_t N, int32_t n_img, int32_t S, int32_t d, int32_t heads,
                                        void* stream) {
  const int beams = N / n_img;
  if (beams != 3 || d / heads != 64) return ODIC_EUNSUPPORTED;
  const size_t sh = (size_t)(3 * S + 256 * 3 + 3) * sizeof(float);
#define ODIC_DBG_X(V)                                                                                                  \
  hipLaunchKernelGGL((cross_attn_step_kernel<3, V>), dim3(n_img, heads, 1), dim3(256), sh, (hipStream_t)stream, q,       \
                     (long)ldq, kv, (long)ldkv, koff, voff, enc_len, row_valid, out, (long)ldo, beams, S, d, heads)
  switch (variant) {
    case 0: ODIC_DBG_X(0); break;
    case 1: ODIC_DBG_X(1); break;
    case 2: ODIC_DBG_X(2); break;
    case 3: ODIC_DBG_X(3); break;
    case 4: ODIC_DBG_X(4); break;
    default: return ODIC_EINVAL;
  }
#undef ODIC_DBG_X
  return odic_launch_status();
}
// state: int32[16] from the caller ({count, first_site = -1, row, col, snapped, launches, per_variant[10]});
// snap: fp32 [3·beams·d + 2·S·d]; variant = which form `b` came from (the snapshot is taken for variant 1 only)
extern "C" int odic_dbg_compare_snapshot(const float* a, const float* b, int64_t ldo, const float* q, int64_t ldq,
                                         const float* kv, int64_t ldkv, int32_t koff, int32_t voff, int32_t N, int32_t n_img,
                                         int32_t S, int32_t d, int32_t site, void* state, float* snap, int32_t variant,
                                         void* stream) {
  const int n = N * d;
  if (variant < 1 || variant > 9) return ODIC_EINVAL;
  hipLaunchKernelGGL(dbg_compare_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, b, n, (int)ldo, d,
                     site, (DbgState*)state, variant);
  if (variant == 1)
    hipLaunchKernelGGL(dbg_snapshot_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, q, (long)ldq, kv, (long)ldkv, koff,
                       voff, a, b, (long)ldo, N / n_img, S, d, site, (DbgState*)state, snap);
  return odic_launch_status();
}
#endif  // ODIC_XATTN_VARIANTS
