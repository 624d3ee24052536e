// attention.hip — masked cross attention of the mask decoder: 100 queries x up to 12 800 sampled voxels per scene,
// 8 heads of 16 channels (reference models/mask3d.py:547-605 CrossAttentionLayer -> nn.MultiheadAttention with a
// boolean memory_mask).  softmax(q k^T / sqrt(hd) + mask) v without materialising the [heads, queries, keys] score
// tensor: the library path is two skinny batched GEMMs (hd = 16), a 41 MB softmax and a 41 MB float mask per call.
//   forward : split over keys (flash-decoding style partial (o, m, l) per split) + a combine kernel
//   backward: recomputes the probabilities from the saved log-sum-exp (D = rowsum(dO * o) in the prologue); dk / dv
//             per key chunk, dq as partial sums per (split, wave) reduced in a fixed order (deterministic, no float
//             atomics); the mask packed by the forward call is reused
// All matrix products run on v_mfma_f32_32x32x2_f32 with the key / query index as the 32-wide tile dimension.
// Tensors keep the module's sequence-first layout [len, batch, heads*hd]; the mask is the decoder's own
// bool[batch, keys, queries] (True = masked), shared by all heads and packed to bits by a pre-pass.
#include "common.h"

namespace usc {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
__device__ inline int arow(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

constexpr int HD = 16;        // channels per head
constexpr int kMaxL = 128;    // queries

struct AttnParams {
  const float* q;   // [L, B, E]
  const float* k;   // [S, B, E]
  const float* v;   // [S, B, E]
  const uint32_t* mbits;  // [B, S, 4]  bit (query) set = masked
  int L, S, B, H, E;
  int nsplit, keys_per_split;   // keys_per_split multiple of 32
  float scale;
  // forward
  float* o_part;    // [B*H, nsplit, kMaxL, HD]
  float* ml_part;   // [B*H, nsplit, 2, kMaxL]
  float* o;         // [L, B, E]
  float* lse;       // [B*H, kMaxL]
  // backward
  const float* dO;  // [L, B, E]
  const float* O;   // [L, B, E] forward output (backward: D = rowsum(dO * O) in the prologue)
  const float* D;   // [B*H, kMaxL] (unused since the prologue computes it)
  float* dq_part;   // [B*H, nsplit*4, kMaxL, HD]
  float* dq;        // [L, B, E]
  float* dk;        // [S, B, E]
  float* dv;        // [S, B, E]
};

// mask bool[B, S, L] -> bits [B, S, 4]
__global__ __launch_bounds__(256) void mask_pack_kernel(const uint8_t* __restrict__ m, int64_t rows, int L,
                                                       uint32_t* __restrict__ bits) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    const int qi = w * 64 + lane;
    const bool f = qi < L && m[r * L + qi] != 0;
    const unsigned long long b = __ballot(f);
    if (lane == 0) { bits[r * 4 + 2 * w] = (uint32_t)b; bits[r * 4 + 2 * w + 1] = (uint32_t)(b >> 32); }
  }
}

// ---- forward, one key split per workgroup; wave w = query tile w
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
  const int bh = blockIdx.x, b = bh / p.H, hh = bh % p.H, split = blockIdx.y;
  const int q0 = wave * 32;
  if (q0 >= p.L) return;
  const int64_t rs = (int64_t)p.B * p.E;                  // row stride of the [len, B, E] tensors
  const int64_t hoff = (int64_t)b * p.E + hh * HD;
  const bool qok = q0 + i < p.L;
  float qv[8];
  {
    const float* qp = p.q + (int64_t)(qok ? q0 + i : 0) * rs + hoff + 8 * h;
    const float4 a = *reinterpret_cast<const float4*>(qp), c = *reinterpret_cast<const float4*>(qp + 4);
    const float t[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
    for (int u = 0; u < 8; ++u) qv[u] = qok ? t[u] * p.scale : 0.f;
  }
  f32x16 oT;
#pragma unroll
  for (int r = 0; r < 16; ++r) oT[r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int kbeg = split * p.keys_per_split;
  const int kend = kbeg + p.keys_per_split < p.S ? kbeg + p.keys_per_split : p.S;
  // operands of a chunk of 32 keys: k rows (A operand of the scores), v columns (A operand of P.V), mask words;
  // fetched one chunk ahead of the matrix-core work (one wave per SIMD: nothing else hides the latency)
  struct Chunk { float kv[8]; float vT[16]; uint32_t mw[16]; };
  auto load_chunk = [&](int key0, Chunk& c) __attribute__((always_inline)) {
    const int key = key0 + i;
    const float* kp = p.k + (int64_t)(key < p.S ? key : 0) * rs + hoff + 8 * h;
    const float4 ka = *reinterpret_cast<const float4*>(kp), kc = *reinterpret_cast<const float4*>(kp + 4);
    c.kv[0] = ka.x; c.kv[1] = ka.y; c.kv[2] = ka.z; c.kv[3] = ka.w; c.kv[4] = kc.x; c.kv[5] = kc.y; c.kv[6] = kc.z; c.kv[7] = kc.w;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int kr = key0 + arow(t, h);
      const int krc = kr < p.S ? kr : p.S - 1;
      c.vT[t] = i < HD ? p.v[(int64_t)krc * rs + hoff + i] : 0.f;                  // A[hd row i][key(t,h)]
      c.mw[t] = p.mbits[((int64_t)b * p.S + krc) * 4 + wave];
    }
  };
  Chunk cur, nxt;
  if (kbeg < kend) load_chunk(kbeg, cur);
  for (int key0 = kbeg; key0 < kend; key0 += 32) {
    if (key0 + 32 < kend) load_chunk(key0 + 32, nxt);
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) s = MFMA32(cur.kv[t], qv[t], s);       // [keys x queries]
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kr = key0 + arow(r, h);
      const bool masked = kr >= kend || !qok || ((cur.mw[r] >> i) & 1u);
      s[r] = masked ? -INFINITY : s[r];
      mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float m_safe = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __expf(m_run - m_safe);            // m_run = -inf -> 0
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = __expf(s[r] - m_safe); psum += s[r]; }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int r = 0; r < 16; ++r) oT[r] *= alpha;
#pragma unroll
    for (int t = 0; t < 16; ++t) oT = MFMA32(cur.vT[t], s[t], oT);    // masked / out-of-range keys have p = 0
    m_run = m_new;
    cur = nxt;
  }
  l_run += __shfl_xor(l_run, 32, 64);
  if (qok) {
    float* op = p.o_part + (((int64_t)bh * p.nsplit + split) * kMaxL + q0 + i) * HD;
#pragma unroll
    for (int r = 0; r < 8; ++r) op[arow(r, h)] = oT[r];    // regs 0..7 are the hd rows < 16
    if (h == 0) {
      float* ml = p.ml_part + ((int64_t)bh * p.nsplit + split) * 2 * kMaxL;
      ml[q0 + i] = m_run;
      ml[kMaxL + q0 + i] = l_run;
    }
  }
}

// one wave per (bh, query): lane = (split group g = lane >> 4, channel d = lane & 15); every lane walks a quarter of the
// splits (max, then the weighted sums), the four groups are combined by two shuffles in a fixed order.  (The first
// version had one thread per (bh, query, channel) walk all <= 64 splits twice: 16 us at 12 800 keys.)
__global__ __launch_bounds__(256) void attn_combine_kernel(AttnParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int qi = (int)(r % p.L);
  const int bh = (int)(r / p.L);
  if (bh >= p.B * p.H) return;
  const int d = lane & 15, g = lane >> 4;
  float M = -INFINITY;
  for (int s = g; s < p.nsplit; s += 4) M = fmaxf(M, p.ml_part[((int64_t)bh * p.nsplit + s) * 2 * kMaxL + qi]);
  M = fmaxf(M, __shfl_xor(M, 16, 64));
  M = fmaxf(M, __shfl_xor(M, 32, 64));
  const float Ms = M == -INFINITY ? 0.f : M;
  float Lsum = 0.f, acc = 0.f;
  for (int s = g; s < p.nsplit; s += 4) {
    const float* ml = p.ml_part + ((int64_t)bh * p.nsplit + s) * 2 * kMaxL;
    const float w = __expf(ml[qi] - Ms);
    Lsum += ml[kMaxL + qi] * w;
    acc += p.o_part[(((int64_t)bh * p.nsplit + s) * kMaxL + qi) * HD + d] * w;
  }
  Lsum += __shfl_xor(Lsum, 16, 64);
  Lsum += __shfl_xor(Lsum, 32, 64);
  acc += __shfl_xor(acc, 16, 64);
  acc += __shfl_xor(acc, 32, 64);
  if (g == 0) {
    const int b = bh / p.H, hh = bh % p.H;
    p.o[(int64_t)qi * p.B * p.E + (int64_t)b * p.E + hh * HD + d] = Lsum > 0.f ? acc / Lsum : 0.f;
    if (d == 0) p.lse[(int64_t)bh * kMaxL + qi] = Lsum > 0.f ? Ms + __logf(Lsum) : INFINITY;
  }
}

// ---- backward: workgroup = (bh, key split); wave w takes the key chunks w, w+4, ... of the split
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnParams p) {
  __shared__ float sq[kMaxL][HD + 1], sdo[kMaxL][HD + 1];   // q * scale, dO of this (batch, head)
  __shared__ float slse[kMaxL], sD[kMaxL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
  const int bh = blockIdx.x, b = bh / p.H, hh = bh % p.H, split = blockIdx.y;
  const int64_t rs = (int64_t)p.B * p.E;
  const int64_t hoff = (int64_t)b * p.E + hh * HD;
  for (int e = threadIdx.x; e < kMaxL * HD; e += 256) {
    const int qi = e / HD, d = e % HD;
    const bool ok = qi < p.L;
    sq[qi][d] = ok ? p.q[(int64_t)qi * rs + hoff + d] * p.scale : 0.f;
    sdo[qi][d] = ok ? p.dO[(int64_t)qi * rs + hoff + d] : 0.f;
  }
  for (int e = threadIdx.x; e < kMaxL; e += 256) {
    float dsum = 0.f;                     // D = rowsum(dO * O): 16 products per query, recomputed by every workgroup
    if (e < p.L) {                        // (a separate launch + a round trip through memory before)
      const float* dop = p.dO + (int64_t)e * rs + hoff;
      const float* op = p.O + (int64_t)e * rs + hoff;
#pragma unroll
      for (int d = 0; d < HD; ++d) dsum += dop[d] * op[d];
    }
    slse[e] = e < p.L ? p.lse[(int64_t)bh * kMaxL + e] : INFINITY;
    sD[e] = dsum;
  }
  __syncthreads();
  const int ntile = (p.L + 31) >> 5;
  f32x16 dqT[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqT[t][r] = 0.f;
  const int kbeg = split * p.keys_per_split;
  const int kend = kbeg + p.keys_per_split < p.S ? kbeg + p.keys_per_split : p.S;
  struct Chunk { float kv[8]; float vv[8]; float kT[16]; uint32_t mrow[4]; };
  auto load_chunk = [&](int key0, Chunk& c) __attribute__((always_inline)) {
    const int key = key0 + i;
    const bool kok = key < kend;
    const float* kp = p.k + (int64_t)(kok ? key : 0) * rs + hoff + 8 * h;
    const float* vp = p.v + (int64_t)(kok ? key : 0) * rs + hoff + 8 * h;
    const float4 a = *reinterpret_cast<const float4*>(kp), cc = *reinterpret_cast<const float4*>(kp + 4);
    const float4 e = *reinterpret_cast<const float4*>(vp), g = *reinterpret_cast<const float4*>(vp + 4);
    const float t1[8] = {a.x, a.y, a.z, a.w, cc.x, cc.y, cc.z, cc.w}, t2[8] = {e.x, e.y, e.z, e.w, g.x, g.y, g.z, g.w};
#pragma unroll
    for (int u = 0; u < 8; ++u) { c.kv[u] = kok ? t1[u] : 0.f; c.vv[u] = kok ? t2[u] : 0.f; }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int kr = key0 + arow(t, h);
      c.kT[t] = (i < HD && kr < kend) ? p.k[(int64_t)kr * rs + hoff + i] : 0.f;     // A[hd row i][key(t,h)]
    }
    // mask bits of key `key` over the queries (lane = key); out-of-range keys are fully masked
    const uint4 mw = kok ? *reinterpret_cast<const uint4*>(p.mbits + ((int64_t)b * p.S + key) * 4) : make_uint4(~0u, ~0u, ~0u, ~0u);
    c.mrow[0] = mw.x; c.mrow[1] = mw.y; c.mrow[2] = mw.z; c.mrow[3] = mw.w;
  };
  Chunk cur, nxt;
  if (kbeg + 32 * wave < kend) load_chunk(kbeg + 32 * wave, cur);
  for (int key0 = kbeg + 32 * wave; key0 < kend; key0 += 128) {
    if (key0 + 128 < kend) load_chunk(key0 + 128, nxt);
    const int key = key0 + i;
    const bool kok = key < kend;
    const float (&kv)[8] = cur.kv;
    const float (&vv)[8] = cur.vv;
    const float (&kT)[16] = cur.kT;
    const uint32_t (&mrow)[4] = cur.mrow;
    f32x16 dkT, dvT;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkT[r] = 0.f; dvT[r] = 0.f; }
#pragma unroll
    for (int tile = 0; tile < 4; ++tile) {
      if (tile >= ntile) break;
      const int q0 = tile * 32;
      float qv[8], dov[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { qv[u] = sq[q0 + i][8 * h + u]; dov[u] = sdo[q0 + i][8 * h + u]; }
      // ---- S1 orientation [keys x queries]: dq
      {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int t = 0; t < 8; ++t) { s = MFMA32(kv[t], qv[t], s); dp = MFMA32(vv[t], dov[t], dp); }
        const float lse_j = slse[q0 + i], D_j = sD[q0 + i];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          // the word of key arow(r,h) for this query tile sits in lane arow(r,h) (out-of-range keys: all ones)
          const uint32_t mw = __shfl(mrow[tile], arow(r, h), 64);
          const bool masked = q0 + i >= p.L || ((mw >> i) & 1u);
          const float pr = masked ? 0.f : __expf(s[r] - lse_j);
          s[r] = pr * (dp[r] - D_j) * p.scale;                     // ds^T
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) dqT[tile] = MFMA32(kT[t], s[t], dqT[tile]);
      }
      // ---- S2 orientation [queries x keys]: dk, dv
      {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int t = 0; t < 8; ++t) { s = MFMA32(qv[t], kv[t], s); dp = MFMA32(dov[t], vv[t], dp); }
        f32x16 ds;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qr = q0 + arow(r, h);
          const bool masked = !kok || qr >= p.L || ((mrow[tile] >> arow(r, h)) & 1u);
          const float pr = masked ? 0.f : __expf(s[r] - slse[qr]);
          s[r] = pr;                                               // p
          ds[r] = pr * (dp[r] - sD[qr]);                           // ds (the 1/sqrt(hd) is already in sq)
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int qr = q0 + arow(t, h);
          const float doT = i < HD ? sdo[qr][i] : 0.f;             // A[hd row i][query(t,h)]
          const float qT = i < HD ? sq[qr][i] : 0.f;
          dvT = MFMA32(doT, s[t], dvT);
          dkT = MFMA32(qT, ds[t], dkT);
        }
      }
    }
    if (kok) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        p.dk[(int64_t)key * rs + hoff + arow(r, h)] = dkT[r];
        p.dv[(int64_t)key * rs + hoff + arow(r, h)] = dvT[r];
      }
    }
    cur = nxt;
  }
  // partial dq of this (split, wave)
  float* dst = p.dq_part + (((int64_t)bh * p.nsplit + split) * 4 + wave) * kMaxL * HD;
#pragma unroll
  for (int tile = 0; tile < 4; ++tile) {
    const int qi = tile * 32 + i;
    if (qi < p.L) {
#pragma unroll
      for (int r = 0; r < 8; ++r) dst[qi * HD + arow(r, h)] = dqT[tile][r];
    }
  }
}

// one wave per (bh, query): lane = (partial group jj = lane >> 4, channel d = lane & 15); every lane sums a quarter of the
// (split, wave) partials, the four groups are combined by two shuffles — fixed order, a quarter of the dependent loads
__global__ __launch_bounds__(256) void attn_dq_reduce_kernel(AttnParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int qi = (int)(r % p.L);
  const int bh = (int)(r / p.L);
  if (bh >= p.B * p.H) return;
  const int d = lane & 15, jj = lane >> 4;
  const int n = p.nsplit * 4;
  float s = 0.f;
  for (int j = jj; j < n; j += 4) s += p.dq_part[(((int64_t)bh * n) + j) * kMaxL * HD + qi * HD + d];
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  if (jj == 0) {
    const int b = bh / p.H, hh = bh % p.H;
    p.dq[(int64_t)qi * p.B * p.E + (int64_t)b * p.E + hh * HD + d] = s;
  }
}

// ---------------------------------------------------------------------------------------------------
// Self attention of the 100 queries (reference models/mask3d.py:491-545 SelfAttentionLayer): S = L <= 128 keys, no
// mask.  One launch each way; no partial sum leaves a workgroup, every sum has a fixed order (no float atomics):
// bit-reproducible at any load.
//   forward : wave w = query tile w, the <= 4 key chunks in sequence (online softmax), o and lse written directly
//   backward: see self_attn_bwd_kernel (D = rowsum(dO * o) in the prologue; partial tiles summed through LDS)
__global__ __launch_bounds__(256) void self_attn_fwd_kernel(AttnParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
  const int bh = blockIdx.x, b = bh / p.H, hh = bh % p.H;
  const int q0 = wave * 32;
  if (q0 >= p.L) return;
  const int64_t rs = (int64_t)p.B * p.E;
  const int64_t hoff = (int64_t)b * p.E + hh * HD;
  const bool qok = q0 + i < p.L;
  float qv[8];
  {
    const float* qp = p.q + (int64_t)(qok ? q0 + i : 0) * rs + hoff + 8 * h;
    const float4 a = *reinterpret_cast<const float4*>(qp), c = *reinterpret_cast<const float4*>(qp + 4);
    const float t[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
    for (int u = 0; u < 8; ++u) qv[u] = qok ? t[u] * p.scale : 0.f;
  }
  f32x16 oT;
#pragma unroll
  for (int r = 0; r < 16; ++r) oT[r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  for (int key0 = 0; key0 < p.S; key0 += 32) {
    float kv[8], vT[16];
    {
      const int key = key0 + i;
      const float* kp = p.k + (int64_t)(key < p.S ? key : 0) * rs + hoff + 8 * h;
      const float4 ka = *reinterpret_cast<const float4*>(kp), kc = *reinterpret_cast<const float4*>(kp + 4);
      kv[0] = ka.x; kv[1] = ka.y; kv[2] = ka.z; kv[3] = ka.w; kv[4] = kc.x; kv[5] = kc.y; kv[6] = kc.z; kv[7] = kc.w;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int kr = key0 + arow(t, h);
        vT[t] = (i < HD && kr < p.S) ? p.v[(int64_t)kr * rs + hoff + i] : 0.f;
      }
    }
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) s = MFMA32(kv[t], qv[t], s);           // [keys x queries]
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool masked = key0 + arow(r, h) >= p.S || !qok;
      s[r] = masked ? -INFINITY : s[r];
      mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float m_safe = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __expf(m_run - m_safe);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = __expf(s[r] - m_safe); psum += s[r]; }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int r = 0; r < 16; ++r) oT[r] *= alpha;
#pragma unroll
    for (int t = 0; t < 16; ++t) oT = MFMA32(vT[t], s[t], oT);
    m_run = m_new;
  }
  l_run += __shfl_xor(l_run, 32, 64);
  if (qok) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    float* op = p.o + (int64_t)(q0 + i) * rs + hoff;
#pragma unroll
    for (int r = 0; r < 8; ++r) op[arow(r, h)] = oT[r] * inv;
    if (h == 0) p.lse[(int64_t)bh * kMaxL + q0 + i] = l_run > 0.f ? m_run + __logf(l_run) : INFINITY;
  }
}

// backward, grid (batch*head, 2 * query tiles): the first half of blockIdx.y owns the dq of one query tile (wave w =
// key chunk w), the second half the dk / dv of one key chunk (wave w = query tile w); the four waves' partial tiles
// are summed through LDS in wave order.  s / dp are recomputed by both kinds of workgroup (16 MFMAs) so that no
// partial leaves a workgroup: 64 short workgroups instead of 8 long ones (28 -> ~8 us at 100 queries).
__global__ __launch_bounds__(256) void self_attn_bwd_kernel(AttnParams p, const float* __restrict__ o) {
  __shared__ float sq[kMaxL][HD + 1], sdo[kMaxL][HD + 1];   // q * scale, dO of this (batch, head)
  __shared__ float slse[kMaxL], sD[kMaxL];
  __shared__ float red[2][4][32][HD];                       // per-wave partial tiles ([0]: dq or dk, [1]: dv)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
  const int bh = blockIdx.x, b = bh / p.H, hh = bh % p.H;
  const int ntile = (p.L + 31) >> 5;
  const bool dq_block = (int)blockIdx.y < ntile;
  const int own = dq_block ? blockIdx.y : blockIdx.y - ntile;      // query tile (dq) or key chunk (dk, dv) owned
  const int64_t rs = (int64_t)p.B * p.E;
  const int64_t hoff = (int64_t)b * p.E + hh * HD;
  for (int e = threadIdx.x; e < kMaxL * HD; e += 256) {
    const int qi = e / HD, d = e % HD;
    const bool ok = qi < p.L;
    sq[qi][d] = ok ? p.q[(int64_t)qi * rs + hoff + d] * p.scale : 0.f;
    sdo[qi][d] = ok ? p.dO[(int64_t)qi * rs + hoff + d] : 0.f;
  }
  for (int e = threadIdx.x; e < kMaxL; e += 256) {
    float dsum = 0.f;
    if (e < p.L) {
      const float* dop = p.dO + (int64_t)e * rs + hoff;
      const float* op = o + (int64_t)e * rs + hoff;
#pragma unroll
      for (int d = 0; d < HD; ++d) dsum += dop[d] * op[d];
    }
    slse[e] = e < p.L ? p.lse[(int64_t)bh * kMaxL + e] : INFINITY;
    sD[e] = dsum;
  }
  __syncthreads();
  const int q0 = (dq_block ? own : wave) * 32;             // this wave's query tile
  const int key0 = (dq_block ? wave : own) * 32;           // this wave's key chunk
  const bool active = q0 < p.L && key0 < p.S;
  f32x16 accA, accB;
#pragma unroll
  for (int r = 0; r < 16; ++r) { accA[r] = 0.f; accB[r] = 0.f; }
  if (active) {
    const int key = key0 + i;
    const bool kok = key < p.S;
    float kv[8], vv[8];
    {
      const float* kp = p.k + (int64_t)(kok ? key : 0) * rs + hoff + 8 * h;
      const float* vp = p.v + (int64_t)(kok ? key : 0) * rs + hoff + 8 * h;
      const float4 a = *reinterpret_cast<const float4*>(kp), cc = *reinterpret_cast<const float4*>(kp + 4);
      const float4 e = *reinterpret_cast<const float4*>(vp), g = *reinterpret_cast<const float4*>(vp + 4);
      const float t1[8] = {a.x, a.y, a.z, a.w, cc.x, cc.y, cc.z, cc.w}, t2[8] = {e.x, e.y, e.z, e.w, g.x, g.y, g.z, g.w};
#pragma unroll
      for (int u = 0; u < 8; ++u) { kv[u] = kok ? t1[u] : 0.f; vv[u] = kok ? t2[u] : 0.f; }
    }
    float qv[8], dov[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { qv[u] = sq[q0 + i][8 * h + u]; dov[u] = sdo[q0 + i][8 * h + u]; }
    if (dq_block) {   // [keys x queries]: ds^T, then dq^T[hd][query] += k^T ds^T
      f32x16 sT, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sT[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int t = 0; t < 8; ++t) { sT = MFMA32(kv[t], qv[t], sT); dp = MFMA32(vv[t], dov[t], dp); }
      const float lse_j = slse[q0 + i], D_j = sD[q0 + i];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool masked = q0 + i >= p.L || key0 + arow(r, h) >= p.S;
        const float pr = masked ? 0.f : __expf(sT[r] - lse_j);
        sT[r] = pr * (dp[r] - D_j) * p.scale;
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int kr = key0 + arow(t, h);
        const float kT = (i < HD && kr < p.S) ? p.k[(int64_t)kr * rs + hoff + i] : 0.f;     // A[hd row i][key(t,h)]
        accA = MFMA32(kT, sT[t], accA);
      }
    } else {          // [queries x keys]: p and ds, then dv^T[hd][key] += dO^T p, dk^T[hd][key] += q^T ds
      f32x16 sm, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sm[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int t = 0; t < 8; ++t) { sm = MFMA32(qv[t], kv[t], sm); dp = MFMA32(dov[t], vv[t], dp); }
      f32x16 ds;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qr = q0 + arow(r, h);
        const bool masked = !kok || qr >= p.L;
        const float pr = masked ? 0.f : __expf(sm[r] - slse[qr]);
        sm[r] = pr;
        ds[r] = pr * (dp[r] - sD[qr]);
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int qr = q0 + arow(t, h);
        const float doT = i < HD ? sdo[qr][i] : 0.f;
        const float qT = i < HD ? sq[qr][i] : 0.f;
        accB = MFMA32(doT, sm[t], accB);
        accA = MFMA32(qT, ds[t], accA);
      }
    }
  }
  // C[hd row][col i]: registers 0..7 hold the hd rows < 16
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    red[0][wave][i][arow(r, h)] = accA[r];
    red[1][wave][i][arow(r, h)] = accB[r];
  }
  __syncthreads();
  const int nw = dq_block ? (p.S + 31) >> 5 : ntile;       // waves that held a real partial; the rest wrote zeros
  const int row0 = own * 32;                               // first query (dq) / key (dk, dv) row of this workgroup
  for (int e = threadIdx.x; e < 32 * HD; e += 256) {
    const int rr = e / HD, d = e % HD;
    const int row = row0 + rr;
    if (row >= p.L) continue;                              // (S = L)
    float a0 = red[0][0][rr][d], a1 = red[1][0][rr][d];
    for (int w = 1; w < nw; ++w) { a0 += red[0][w][rr][d]; a1 += red[1][w][rr][d]; }
    if (dq_block) {
      p.dq[(int64_t)row * rs + hoff + d] = a0;
    } else {
      p.dk[(int64_t)row * rs + hoff + d] = a0;
      p.dv[(int64_t)row * rs + hoff + d] = a1;
    }
  }
}

static int pick_splits(int BH, int S) {
  int chunks = (S + 31) / 32;
  int want = (1024 + BH * 4 - 1) / (BH * 4);     // ~1024 waves in flight
  if (want > chunks) want = chunks;
  if (want > 64) want = 64;
  if (want < 1) want = 1;
  return want;
}

}  // namespace
}  // namespace usc

using namespace usc;

extern "C" {

int64_t usc_attn_ws_bytes(int32_t L, int32_t S, int32_t B, int32_t H) {
  (void)L;
  const int64_t BH = (int64_t)B * H;
  const int ns = pick_splits((int)BH, S);
  const int64_t bits = align_up((int64_t)B * S * 4 * 4, 256);
  const int64_t part = BH * ns * 4 * kMaxL * HD * 4;            // max(o_part, dq_part)
  const int64_t ml = BH * ns * 2 * kMaxL * 4;
  return bits + part + ml + 256;
}

/* forward: o [L,B,E], lse [B*H,128] (saved for the backward) */
int usc_attn_fwd(const float* q, const float* k, const float* v, const uint8_t* mask, int32_t L, int32_t S, int32_t B,
                 int32_t H, int32_t E, float* o, float* lse, void* ws, int64_t ws_bytes, usc_stream_t s) {
  USC_REQUIRE(L >= 1 && L <= kMaxL && S >= 1 && B >= 1 && H >= 1 && E == H * HD,
              "usc_attn_fwd: needs head dim 16 and at most 128 queries");
  USC_REQUIRE(q && k && v && mask && o && lse && ws && ws_bytes >= usc_attn_ws_bytes(L, S, B, H), "usc_attn_fwd: bad argument");
  hipStream_t st = as_stream(s);
  AttnParams p{};
  p.q = q; p.k = k; p.v = v; p.L = L; p.S = S; p.B = B; p.H = H; p.E = E; p.scale = 1.0f / sqrtf((float)HD);
  p.nsplit = pick_splits(B * H, S);
  p.keys_per_split = (int)align_up(ceil_div(S, p.nsplit), 32);
  char* w = (char*)ws;
  uint32_t* bits = (uint32_t*)w;
  w += align_up((int64_t)B * S * 4 * 4, 256);
  p.mbits = bits;
  p.o_part = (float*)w;
  w += (int64_t)B * H * p.nsplit * 4 * kMaxL * HD * 4;
  p.ml_part = (float*)w;
  p.o = o; p.lse = lse;
  hipLaunchKernelGGL(mask_pack_kernel, dim3((unsigned)ceil_div((int64_t)B * S, 4)), dim3(256), 0, st, mask, (int64_t)B * S, (int)L, bits);
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(B * H, p.nsplit), dim3(256), 0, st, p);
  hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)ceil_div((int64_t)B * H * L, 4)), dim3(256), 0, st, p);
  USC_CHECK_LAUNCH("usc_attn_fwd");
  return USC_OK;
}

/* backward: dq [L,B,E], dk / dv [S,B,E]; D scratch f32[B*H,128] */
int usc_attn_bwd(const float* q, const float* k, const float* v, const uint8_t* mask, const float* o, const float* lse,
                 const float* dO, int32_t L, int32_t S, int32_t B, int32_t H, int32_t E, float* dq, float* dk, float* dv,
                 int32_t mask_bits_in_ws, void* ws, int64_t ws_bytes, usc_stream_t s) {
  USC_REQUIRE(L >= 1 && L <= kMaxL && S >= 1 && B >= 1 && H >= 1 && E == H * HD,
              "usc_attn_bwd: needs head dim 16 and at most 128 queries");
  USC_REQUIRE(q && k && v && mask && o && lse && dO && dq && dk && dv && ws && ws_bytes >= usc_attn_ws_bytes(L, S, B, H),
              "usc_attn_bwd: bad argument");
  hipStream_t st = as_stream(s);
  AttnParams p{};
  p.q = q; p.k = k; p.v = v; p.L = L; p.S = S; p.B = B; p.H = H; p.E = E; p.scale = 1.0f / sqrtf((float)HD);
  p.nsplit = pick_splits(B * H, S);
  p.keys_per_split = (int)align_up(ceil_div(S, p.nsplit), 32);
  char* w = (char*)ws;
  uint32_t* bits = (uint32_t*)w;
  w += align_up((int64_t)B * S * 4 * 4, 256);
  p.mbits = bits;
  p.dq_part = (float*)w;
  w += (int64_t)B * H * p.nsplit * 4 * kMaxL * HD * 4;
  p.D = nullptr; p.O = o; p.lse = (float*)lse; p.dO = dO; p.dq = dq; p.dk = dk; p.dv = dv;
  // mask_bits_in_ws: `ws` is the forward call's workspace and still holds the packed mask at its start (the forward's
  // partial sums behind it are dead: this call overwrites them)
  if (!mask_bits_in_ws)
    hipLaunchKernelGGL(mask_pack_kernel, dim3((unsigned)ceil_div((int64_t)B * S, 4)), dim3(256), 0, st, mask, (int64_t)B * S, (int)L, bits);
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(B * H, p.nsplit), dim3(256), 0, st, p);
  hipLaunchKernelGGL(attn_dq_reduce_kernel, dim3((unsigned)ceil_div((int64_t)B * H * L, 4)), dim3(256), 0, st, p);
  USC_CHECK_LAUNCH("usc_attn_bwd");
  return USC_OK;
}

/* self attention (no mask, S = L <= 128): o [L,B,E], lse [B*H,128]; one launch */
int usc_self_attn_fwd(const float* q, const float* k, const float* v, int32_t L, int32_t B, int32_t H, int32_t E,
                      float* o, float* lse, usc_stream_t s) {
  USC_REQUIRE(L >= 1 && L <= kMaxL && B >= 1 && H >= 1 && E == H * HD,
              "usc_self_attn_fwd: needs head dim 16 and at most 128 queries");
  USC_REQUIRE(q && k && v && o && lse, "usc_self_attn_fwd: bad argument");
  AttnParams p{};
  p.q = q; p.k = k; p.v = v; p.L = L; p.S = L; p.B = B; p.H = H; p.E = E; p.scale = 1.0f / sqrtf((float)HD);
  p.o = o; p.lse = lse;
  hipLaunchKernelGGL(self_attn_fwd_kernel, dim3(B * H), dim3(256), 0, as_stream(s), p);
  USC_CHECK_LAUNCH("usc_self_attn_fwd");
  return USC_OK;
}

int usc_self_attn_bwd(const float* q, const float* k, const float* v, const float* o, const float* lse, const float* dO,
                      int32_t L, int32_t B, int32_t H, int32_t E, float* dq, float* dk, float* dv, usc_stream_t s) {
  USC_REQUIRE(L >= 1 && L <= kMaxL && B >= 1 && H >= 1 && E == H * HD,
              "usc_self_attn_bwd: needs head dim 16 and at most 128 queries");
  USC_REQUIRE(q && k && v && o && lse && dO && dq && dk && dv, "usc_self_attn_bwd: bad argument");
  AttnParams p{};
  p.q = q; p.k = k; p.v = v; p.L = L; p.S = L; p.B = B; p.H = H; p.E = E; p.scale = 1.0f / sqrtf((float)HD);
  p.lse = (float*)lse; p.dO = dO; p.dq = dq; p.dk = dk; p.dv = dv;
  hipLaunchKernelGGL(self_attn_bwd_kernel, dim3(B * H, 2 * ((L + 31) / 32)), dim3(256), 0, as_stream(s), p, o);
  USC_CHECK_LAUNCH("usc_self_attn_bwd");
  return USC_OK;
}

}  // extern "C"
