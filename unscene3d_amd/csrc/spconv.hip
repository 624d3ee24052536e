// spconv.hip — sparse convolution as rulebook-driven implicit GEMM on the
// gfx950 matrix cores.
//
// Three kernels cover every conv in Res16UNet34C + Mask3D, forward and backward:
//   gather_gemm  (output-stationary over a dense neighbour table nbr[K][N_out]):
//                k3/s1 conv fwd + dgrad, k2/s2 conv fwd, k2/s2 conv-transpose dgrad,
//                1x1 conv / linear (nbr == NULL)
//   pairs_gemm   (one-parent form driven by per-offset pair lists):
//                k2/s2 conv-transpose fwd, k2/s2 conv dgrad
//   wgrad        dW[k] = sum_pairs a^T b, split over pair chunks, ordered reduction
//
// Arithmetic: v_mfma_f32_32x32x2_f32 — exact fp32 (bitwise an fmaf chain), so the
// result is within fp32 round-off of the reference's cuBLAS/CPU GEMMs.
// Each 64-lane wave owns a 32-row x (NB*32)-column output tile with NB 32x32
// accumulators in registers; A rows are gathered straight from HBM/L2 with 16-B
// loads (the reduction index is permuted so that a lane's float4 feeds four
// consecutive MFMA steps), B (the weight slice W[k]) is read as 128-B coalesced
// row segments that stay L1/L2 resident across the waves of a CU.  Offsets with
// no neighbour in the wave's 32 rows are skipped with a wave-uniform ballot.
#include "common.h"

#include <stdlib.h>

#include <vector>

namespace usc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

struct GemmParams {
  const float* in;      // [n_in, cin]
  const float* W;       // [K, cin, cout]
  const int32_t* nbr;   // [K, n_out] or NULL (identity)
  const float* bias;    // [cout] or NULL
  float* out;           // [n_out, cout]  (G == 1)  or partial [G, n_out, cout]
  int64_t n_out;
  int cin, cout, K;
  int accumulate;
  int G;                // offset groups (blockIdx.z); >1 -> partial sums, reduced afterwards
  int TM;               // tile-compacted kernel: output rows per workgroup
  int wt1;              // aligned kernel, K == 1: W is given as [cout][cin] (nn.Linear's layout), B[c][n] = W[n][c]
  // pairs form
  const int32_t* rows_in;
  const int32_t* rows_out;
  const int64_t* koff;
  // launch statistics (usc_launch_stats_begin): {min start, max end (wall clock ticks), real pairs, -} of this launch, or NULL
  unsigned long long* stats;
};

// MFMA C/D layout (32x32): col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
__device__ inline int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

template <int NB> struct BVec;
template <> struct BVec<1> { float v[1]; };
template <> struct BVec<2> { float v[2]; };
template <> struct BVec<3> { float v[3]; };
template <> struct BVec<4> { float v[4]; };

// Column mapping of the aligned kernels: MFMA column j of accumulator nb is the
// actual output column n0 + NB*j + nb, so that a lane's NB B-operands (and its NB
// results per row) are CONTIGUOUS in memory: one NB-dword load per reduction step,
// one NB-dword store per output row.
template <int NB>
__device__ inline void load_b(const float* __restrict__ p, float (&b)[NB]) {
  if (NB == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    b[0] = t.x; b[1 % NB] = t.y; b[2 % NB] = t.z; b[3 % NB] = t.w;
  } else if (NB == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    b[0] = t.x; b[1 % NB] = t.y;
  } else {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) b[nb] = p[nb];
  }
}

// Locate this wave's work item.  Table form: 32 consecutive output rows.  List
// form: 32 consecutive pairs of one offset k (every offset padded to 32 pairs).
template <bool LIST>
__device__ inline bool locate_tile(const GemmParams& p, int wave, int i, int& k_begin, int& k_end,
                                   int64_t& my_out_row, int64_t& my_in_row_list) {
  my_out_row = -1;
  my_in_row_list = -1;
  if (LIST) {
    int64_t t = (int64_t)blockIdx.x * 4 + wave;
    for (int k = 0; k < p.K; ++k) {
      const int64_t b = p.koff[k], e = p.koff[k + 1];
      const int64_t nt = (e - b + 31) >> 5;
      if (t < nt) {
        const int64_t pp = b + t * 32 + i;
        if (pp < e) {
          my_out_row = p.rows_out[pp];
          my_in_row_list = p.rows_in[pp];
        }
        k_begin = k;
        k_end = k + 1;
        return true;
      }
      t -= nt;
    }
    return false;
  } else {
    const int64_t tile_row0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
    if (tile_row0 >= p.n_out) return false;
    const int64_t r = tile_row0 + i;
    my_out_row = (r < p.n_out) ? r : -1;
    // offset group of this block
    const int Kg = (p.K + p.G - 1) / p.G;
    k_begin = blockIdx.z * Kg;
    k_end = k_begin + Kg < p.K ? k_begin + Kg : p.K;
    return true;
  }
}

// Fast path: cin % 32 == 0 and cout % (32*NB) == 0 — no bounds checks inside the loop,
// every load unconditional (invalid rows read row 0 and are zeroed with a select).
template <int NB, bool LIST, bool WT = false>   // WT: K == 1, W given as [cout][cin] (a linear layer's weight), see GemmParams::wt1
__global__ __launch_bounds__(256) void gather_gemm_aligned_kernel(GemmParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.y * (NB * 32);
  const int cin = p.cin, cout = p.cout;
  int k_begin, k_end;
  int64_t my_out_row, my_in_row_list;
  if (!locate_tile<LIST>(p, wave, i, k_begin, k_end, my_out_row, my_in_row_list)) return;

  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  for (int k = k_begin; k < k_end; ++k) {
    int64_t in_row;
    if (LIST) in_row = my_in_row_list;
    else if (p.nbr) in_row = (my_out_row >= 0) ? (int64_t)p.nbr[(int64_t)k * p.n_out + my_out_row] : -1;
    else in_row = my_out_row;
    const bool valid = in_row >= 0;
    if (!__any(valid)) continue;  // wave-uniform: no neighbour at this offset in these 32 rows
    const float* arow = p.in + (valid ? in_row : 0) * (int64_t)cin + 4 * h;
    const float* wk = p.W + (int64_t)k * cin * cout + (int64_t)(4 * h) * cout + n0 + NB * i;

    for (int c0 = 0; c0 < cin; c0 += 32) {
      float4 a[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] = *reinterpret_cast<const float4*>(arow + c0 + 8 * t);
      float b[16][NB];
      if (WT) {
        // a lane's column n0 + NB*i + nb of B is ROW n of W: its four k-steps of a quad are one 16-byte load
        const float* wr = p.W + (int64_t)(n0 + NB * i) * cin + 4 * h + c0;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float4 w4 = *reinterpret_cast<const float4*>(wr + (int64_t)nb * cin + 8 * t);
            b[4 * t + 0][nb] = w4.x; b[4 * t + 1][nb] = w4.y; b[4 * t + 2][nb] = w4.z; b[4 * t + 3][nb] = w4.w;
          }
      } else {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) load_b<NB>(wk + (int64_t)(c0 + 8 * t + j) * cout, b[4 * t + j]);
      }
      if (!valid) {
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float av[4] = {a[t].x, a[t].y, a[t].z, a[t].w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = MFMA32(av[j], b[4 * t + j][nb], acc[nb]);
      }
    }
  }

  float* outp = p.out;
  if (!LIST && p.G > 1) outp += (int64_t)blockIdx.z * p.n_out * cout;
  const bool direct = LIST || p.G == 1;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t orow = __shfl(my_out_row, acc_row(r, h), 64);
    if (orow < 0) continue;
    float v[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) v[nb] = acc[nb][r];
    float* dst = outp + orow * (int64_t)cout + n0 + NB * i;
    if (direct) {
      if (p.bias) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) v[nb] += p.bias[n0 + NB * i + nb];
      }
      if (p.accumulate) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) v[nb] += dst[nb];
      }
    }
    if (NB == 4) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1 % NB], v[2 % NB], v[3 % NB]);
    else if (NB == 2) *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1 % NB]);
    else {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) dst[nb] = v[nb];
    }
  }
}

// ---------------------------------------------------------------------------
// Tile-compacted gather GEMM (large maps).  A workgroup owns TM consecutive output rows and keeps
// their [TM][32*NB] fp32 accumulators in LDS.  For every kernel offset k the rows that actually
// have a neighbour are compacted (ballot + popcount prefix) into pair lists, so the matrix cores
// only see groups of 32 REAL (in,out) pairs instead of 32 rows of which ~half are padding
// (random-order surface voxels: 13 of 27 neighbours present).  The work items (k, group) of a
// tile are dealt round-robin to the 4 waves; each wave gathers its 32 input rows, runs the MFMA
// reduction over Cin into registers and then adds the 32xBN result into the LDS accumulators.
// Flushes are serialised by an LDS ticket in item order, which makes the fp32 summation order —
// and therefore the result — bit-reproducible without float atomics.
constexpr int kMaxK = 27;
constexpr int kMaxItems = kMaxK * 8;   // TM <= 256 -> at most 8 groups of 32 pairs per offset
constexpr int kZeroRowFloats = 4096;
// padding lanes of a pair group gather from this all-zero row instead of selecting zeros after the load
// (a select on a loaded register makes the compiler wait for the load — and everything older — first)
__device__ float g_zero_row[kZeroRowFloats + 8];

#ifdef USC_PHASE_STATS   /* developer build (tools/build_ablate.sh phase "-DUSC_PHASE_STATS"): where a tile's cycles go */
__device__ unsigned long long g_phase[16];
#define USC_PH(idx, val) atomicAdd(&g_phase[idx], (unsigned long long)(val))
#endif
#ifndef USC_COMPACT_WAVES
#define USC_COMPACT_WAVES 8          /* developer builds: 16 = one 1024-thread workgroup per CU (tools/build_ablate.sh) */
#endif
constexpr int kCompactWaves = USC_COMPACT_WAVES;   // 8: 512-thread workgroups, two resident per CU when their LDS fits

template <int NB>
__global__ __launch_bounds__(64 * kCompactWaves, 4) void gather_gemm_compact_kernel(GemmParams p) {
  constexpr int BN = NB * 32;
  constexpr int NT = 64 * kCompactWaves;
  const int TM = p.TM;   // rows per tile (multiple of 4, <= 256), chosen so that tiles fill whole CU rounds
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* accT = reinterpret_cast<float*>(smem);                        // [TM][BN]
  int32_t* pl_in = reinterpret_cast<int32_t*>(accT + (TM + 1) * BN);   // [K][TM]   (accT[TM] = the dummy row of padding pairs)
  int32_t* cnt = pl_in + kMaxK * TM;                                   // [32]
  int32_t* item_start = cnt + 32;                                      // [32]
  volatile int32_t* ticket = item_start + 32;                          // [1] (+3 pad)
  typedef __attribute__((address_space(3))) volatile int32_t lds_vint;
  lds_vint* ticket3 = (lds_vint*)ticket;
  uint8_t* pl_loc = reinterpret_cast<uint8_t*>(item_start + 36);       // [K][TM]
  int32_t* item_desc = reinterpret_cast<int32_t*>(pl_loc + kMaxK * TM);  // [kMaxItems]: pbase | npairs << 16 | k << 24

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.y * BN;
  const int cin = p.cin, cout = p.cout, K = p.K;
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch rule, speed only); give every XCD
  // a CONTIGUOUS range of tiles so that the gathers of neighbouring tiles (which share most of their input
  // rows once the rows are in z-order) hit the same 4 MB L2 instead of eight different ones.
  int64_t tile;
  {
    const int64_t nwg = gridDim.x, bid = blockIdx.x;
    const int64_t q = nwg / 8, r = nwg % 8, xcd = bid % 8, j = bid / 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int64_t r0 = tile * TM;
  if (p.stats && threadIdx.x == 0) atomicMin(p.stats, (unsigned long long)wall_clock64());
#ifdef USC_PHASE_STATS
  const long long ph_t0 = clock64();
  long long ph_wait = 0, ph_flush = 0;
#endif

  // ---- prologue: zero accumulators, compact the neighbour table of this tile per offset
  for (int e = threadIdx.x; e < (TM + 1) * BN / 4; e += NT) reinterpret_cast<float4*>(accT)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (threadIdx.x == 0) *ticket = 0;
  {
    // every neighbour-table word this wave compacts (<= 4 offsets x 4 chunks of 64 rows) is requested before the first one
    // is used: the ballot passes below depend on each other only through `base`, but one load per pass, each waited
    // for, put ~10 dependent HBM round trips in front of every tile (round 6: tools/compact_phase.py, 6 % of a tile)
    constexpr int kOffPerWave = (kMaxK + kCompactWaves - 1) / kCompactWaves;
    int nv[kOffPerWave][4];
#pragma unroll
    for (int j = 0; j < kOffPerWave; ++j) {
      const int k = wave + j * kCompactWaves;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int lr = c * 64 + lane;
        const int64_t row = r0 + lr;
        nv[j][c] = (k < K && lr < TM && row < p.n_out) ? (p.nbr ? p.nbr[(int64_t)k * p.n_out + row] : (int)row) : -1;
      }
    }
#pragma unroll
    for (int j = 0; j < kOffPerWave; ++j) {
      const int k = wave + j * kCompactWaves;
      if (k >= K) break;
      int base = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c * 64 >= TM) break;
        const int v = nv[j][c];
        const bool f = v >= 0;
        const unsigned long long m = __ballot(f);
        if (f) {
          const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
          pl_in[k * TM + pos] = v;
          pl_loc[k * TM + pos] = (uint8_t)(c * 64 + lane);
        }
        base += __popcll(m);
      }
      if (lane == 0) cnt[k] = base;
    }
  }
  __syncthreads();
  if (wave == 0) {
    const int c = (lane < K) ? cnt[lane] : 0;
    const int items = (c + 31) >> 5;
    const int inc = wave_inclusive_scan(items);
    if (lane < 32) item_start[lane] = inc - items;   // exclusive prefix; entries >= K hold the total
    if (lane < K) {
      for (int g = 0; g < items; ++g) {
        const int np = c - g * 32 < 32 ? c - g * 32 : 32;
        item_desc[inc - items + g] = (lane * TM + g * 32) | (np << 16) | (lane << 24);
      }
    }
  }
  __syncthreads();
  const int total_items = item_start[K < 32 ? K : 31] + (K >= 32 ? 0 : 0);
#ifdef USC_PHASE_STATS
  const long long ph_t1 = clock64();
#endif

  // ---- main loop.  A wave walks its items (item = wave, wave + 8, ...); one item = 32 real pairs of one
  // offset k, reduced over Cin in "quads" of 8 input channels (one 16-byte load of the gathered row per lane
  // feeds four MFMA k-steps; each k-step needs NB weight operands = one NB-dword load).
  // The loads run AHEAD of the matrix cores through two small register rings — gathered rows kDA quads
  // ahead (HBM / Infinity-Cache latency), weights kDB quads ahead (L2 latency) — and the rings run across
  // item boundaries, so the next item's first gathers are in flight while this item is flushed.  (The first
  // version issued a quad's loads right before its MFMAs: the ISA showed `s_waitcnt vmcnt(0)` in front of
  // every k-step, i.e. one exposed L2 round trip per 3 MFMAs, MFMA pipe 50 % busy.)
  struct Item {
    const float* arow;  // gathered input row of lane (i,h) (+4h); padding lanes: the zero row
    const float* wk;    // packed weight slice of offset k for this column block (wave-uniform, lives in SGPRs)
    int pbase;          // index of this item's pair 4h in the offset's pair list (flush: local output rows)
    int npairs;         // real pairs of the item (<= 32)
    int item;           // work item index; -1 = past the end (dummy loads, no flush)
    int first;          // index of the first item of this item's offset (= flushes that must have happened before)
  };
  const int nq = cin >> 3;   // quads per item (multiple of 4)
  // this lane's byte offset inside a weight slice: rows 4h.., columns n0 + NB*i..  (constant for the kernel, so
  // every weight load is  SGPR base + this VGPR  and costs no vector address arithmetic)
  const float* wp_cb = p.W + (int64_t)blockIdx.y * K * cin * BN;   // p.W: weights packed by weight_pack_kernel
  auto make_item = [&](int item) -> Item {
    Item st;
    if (item >= total_items) {
      st.item = -1; st.pbase = 0; st.npairs = 0; st.first = 0;
      st.arow = g_zero_row + 4 * h;
      st.wk = wp_cb;
      return st;
    }
    const int d = __builtin_amdgcn_readfirstlane(item_desc[item]);
    const int k = d >> 24, npairs = (d >> 16) & 63, pb0 = d & 0xffff;
    const int pidx = pb0 + i;
    st.pbase = pb0 + 4 * h;
    st.npairs = npairs;
    st.item = item;
    st.first = __builtin_amdgcn_readfirstlane(item_start[k]);
    st.arow = (i < npairs ? p.in + (int64_t)pl_in[pidx] * (int64_t)cin : g_zero_row) + 4 * h;
    st.wk = wp_cb + (int64_t)k * cin * BN;
    return st;
  };
#ifndef USC_KDB
#define USC_KDB 1
#endif
#ifndef USC_KRB
#define USC_KRB 2
#endif
#ifndef USC_KDA
#define USC_KDA 3
#endif
  constexpr int kDA = USC_KDA, kDB = USC_KDB;  // prefetch distances in quads
  constexpr int kRA = 4, kRB = USC_KRB;  // ring sizes (the quad loop is unrolled by 4: static slots)
  float4 ra[kRA];
  float4 rb[kRB][NB];   // [slot][accumulator] -> the 4 k-steps of the quad
  f32x16 acc[NB];
  auto zero_acc = [&]() {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
  };
  auto load_a = [&](float4& dst, const float* src) {
#ifdef USC_ABLATE_A
    dst = make_float4(1.f, 2.f, 3.f, 4.f);
#else
    dst = *reinterpret_cast<const float4*>(src);
#endif
  };
  auto load_bq = [&](float4 (&dst)[NB], const float* wq) {   // wq: uniform address of the quad's packed weights
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#ifdef USC_ABLATE_B
      dst[nb] = make_float4(1.f + nb, 2.f + nb, 3.f + nb, 4.f + nb);
#else
      dst[nb] = *reinterpret_cast<const float4*>(wq + nb * 256 + lane * 4);
#endif
    }
  };
  auto flush = [&](const Item& st) {
#ifdef USC_ABLATE_FLUSH
    float sink = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) sink += acc[nb][r];
    if (sink == 1.2345e-30f) accT[lane] = sink;
    zero_acc();
    return;
#endif
    // ordered flush into the LDS accumulators (ticket == item index).  The ticket lives in LDS and is polled
    // with ds_read (an address_space(3) pointer: the generic `volatile int*` compiled to flat_load sc0 sc1 +
    // s_waitcnt vmcnt(0), draining the prefetch rings at every poll).  LDS executes one wave's instructions in
    // issue order, so the hand-off needs no memory fence — which would also wait for vmcnt(0) — only the
    // compiler barriers and lgkmcnt waits below.
#ifdef USC_PHASE_STATS
    const long long ph_f0 = clock64();
#endif
    // The local output rows of this lane's 16 accumulator rows, read BEFORE the ticket (the pair lists do not change after
    // the prologue): accumulator row r = 4g + q belongs to pair 8g + q + 4h of the item, so a lane's rows are four runs of
    // four consecutive bytes — four aligned word reads, all in flight while lane 0 polls.  (Round 6: the first version read
    // one byte per row inside the ordered section, each behind an `s_waitcnt lgkmcnt(0)` — sixteen dependent LDS round
    // trips under the ticket; tools/compact_phase.py put the section at a fifth of a wave's time.)
    uint32_t loc4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) loc4[g] = *reinterpret_cast<const uint32_t*>(pl_loc + st.pbase + 8 * g);
#ifndef USC_ABLATE_TICKET
    if (lane == 0) {
#ifndef USC_POLL_SLEEP
#define USC_POLL_SLEEP 1
#endif
      // The ticket counts FLUSHED items.  Items are numbered offset-major and the items of one offset touch disjoint
      // output rows (a row has at most one neighbour per offset), so an item only has to wait for the offsets before
      // its own — st.first = index of the first item of its offset — not for its siblings: the ~2 items per offset of a
      // 148-row tile flush side by side, and every output element is still summed over k ascending (bit-identical).
      while (*ticket3 < st.first) __builtin_amdgcn_s_sleep(USC_POLL_SLEEP);
    }
    __builtin_amdgcn_wave_barrier();
#endif
#ifdef USC_PHASE_STATS
    const long long ph_f1 = clock64();
#endif
#ifndef USC_NO_SETPRIO
    // the ticket holder is the workgroup's critical path: without priority its LDS/VALU instructions queue
    // behind the other waves' MFMAs (issue is arbitrated by priority, then age) and the serialised section
    // stretched to ~10k cycles per item — measured: ticket alone +3 %, RMW alone +8 %, both together +77 %.
    __builtin_amdgcn_s_setprio(3);
#endif
    asm volatile("" ::: "memory");
    // Plain LDS read-add-write; the ticket gives this wave exclusive, ordered access (ds_add_f32
    // atomics measured 2x slower for the whole kernel).  No branches: a padding pair's rows go to the dummy row
    // accT[TM] (its accumulators are exact zeros — the pair gathered the all-zero row — but -0 + 0 would flip a sign
    // bit in a real row).  The reads of a batch of rows are issued before the first dependent add so the row updates
    // pipeline instead of paying one LDS round trip each.
#ifndef USC_KFB
#define USC_KFB 4
#endif
    constexpr int kFB = USC_KFB;   // rows per batch (register budget: the load rings stay live across the flush)
#ifdef USC_ABLATE_RMW
    {
      float sink = 0.f;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sink += acc[nb][r];
      if (sink == 1.2345e-30f + (float)loc4[0]) accT[lane] = sink;
    }
#else
#pragma unroll
    for (int part = 0; part < 16 / kFB; ++part) {
      float* dsts[kFB];
      float old[kFB][NB];
#pragma unroll
      for (int q = 0; q < kFB; ++q) {
        const int r = part * kFB + q;
        const int prow = (r & 3) + 8 * (r >> 2);   // pair index inside the item is prow + 4h (MFMA C layout)
        const int lr = (prow + 4 * h < st.npairs) ? (int)((loc4[r >> 2] >> (8 * (r & 3))) & 0xffu) : TM;
        dsts[q] = accT + lr * BN + NB * i;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) old[q][nb] = dsts[q][nb];
      }
#pragma unroll
      for (int q = 0; q < kFB; ++q) {
        const int r = part * kFB + q;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) dsts[q][nb] = old[q][nb] + acc[nb][r];
      }
    }
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS updates are done before the ticket moves
#ifndef USC_ABLATE_TICKET
    __builtin_amdgcn_wave_barrier();
    if (lane == 0)   // ds_add on the LDS address (a generic pointer would compile to a flat atomic and drain the load rings)
      __hip_atomic_fetch_add((__attribute__((address_space(3))) int32_t*)ticket3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
#ifndef USC_NO_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#ifdef USC_PHASE_STATS
    { const long long ph_f2 = clock64(); ph_wait += ph_f1 - ph_f0; ph_flush += ph_f2 - ph_f1; }
#endif
    zero_acc();
  };

  Item cur = make_item(wave);
  Item nxt = make_item(wave + kCompactWaves);
  zero_acc();
  if (cur.item >= 0) {
    // running prefetch cursors: `pa` (per lane) = gathered row position kDA quads ahead, `pb` (uniform) = weight
    // row position kDB quads ahead; both hop to the next item's row / weight slice when they pass the item's end
    const float* pa = cur.arow;
    const float* pb = cur.wk;
    int qa = 0, qb = 0;     // quad index (inside its item) the cursors point at
    auto advance_a = [&]() {
      pa += 8; ++qa;
      if (qa == nq) { pa = nxt.arow; qa = 0; }
    };
    auto advance_b = [&]() {
      pb += NB * 256; ++qb;
      if (qb == nq) { pb = nxt.wk; qb = 0; }
    };
#pragma unroll
    for (int q = 0; q < kDA; ++q) { load_a(ra[q % kRA], pa); advance_a(); }
#pragma unroll
    for (int q = 0; q < kDB; ++q) { load_bq(rb[q % kRB], pb); advance_b(); }
    // ONE loop over groups of four quads (static ring slots); the item switch happens inside it, so the ring
    // registers never have to be copied (and therefore waited for) at item boundaries.
    int q0 = 0;
    while (true) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#ifndef USC_LOAD_B_FIRST
        load_a(ra[(u + kDA) % kRA], pa); advance_a();
        load_bq(rb[(u + kDB) % kRB], pb); advance_b();
#else
        // Developer switch, measured in round 6 and NOT kept: vector loads return in order, and the next quad's first MFMA
        // waits for this quad's weight loads — with the gather issued in front of them it also waits for a row requested
        // only one quad earlier, whatever the ring's nominal distance of kDA quads.  Weights first doubles that distance
        // (the ISA shows vmcnt(7) instead of vmcnt(6)): 0.511 / 0.478 ms against 0.509 / 0.466 (96 -> 96, 148 564 rows,
        // forward / input gradient), 24.1 against 23.9 ms per step, and kDA = 2 is as fast as 3 — the gathered rows'
        // latency is not what this loop waits for.
        load_bq(rb[(u + kDB) % kRB], pb); advance_b();
        load_a(ra[(u + kDA) % kRA], pa); advance_a();
#endif
        __builtin_amdgcn_sched_barrier(0);
        const float4 a4 = ra[u % kRA];
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const float4 b4 = rb[u % kRB][nb];
            const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
            acc[nb] = MFMA32(av[j], bv[j], acc[nb]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      q0 += 4;
      if (q0 >= nq) {
        flush(cur);
        if (nxt.item < 0) break;
        cur = nxt;
        nxt = make_item(cur.item + kCompactWaves);
        q0 = 0;
      }
    }
  }
#ifdef USC_PHASE_STATS
  const long long ph_tw = clock64();
#endif
  __syncthreads();
#ifdef USC_PHASE_STATS
  const long long ph_t3 = clock64();
  if (lane == 0) { USC_PH(3, ph_t3 - ph_tw); USC_PH(4, ph_wait); USC_PH(5, ph_flush); USC_PH(8, ph_tw - ph_t1); }
#endif

  // ---- epilogue: coalesced copy of the tile to HBM
  // (Round 6, measured and not kept: in this loop the compiler waits for `vmcnt(0)` at the join of the bias / accumulate
  //  branches, i.e. for the previous iteration's store — a bias-free loop of its own without that wait, and the prologue's
  //  sixteen table loads made unconditional so that all of them are in flight at once, took the write-back from 47.6 k to
  //  37.1 k and the prologue from 30.1 k to 27.4 k cycles of a 579 k-cycle tile in the phase-counting build, bit-identical
  //  outputs — and the uninstrumented kernel did not move: 490 / 493 vs 501 / 505 us on 96 -> 96 x 148 564 rows, 23.4 / 24.0 /
  //  23.4 vs 23.5 / 24.3 / 23.6 ms per step.  The co-resident workgroup's matrix-core work already covers these phases.  The
  //  same goes for the ordered section as 48 fire-and-forget ds_add_f32 per lane: bit-identical and 2.5x slower, 1 238 us.)
  for (int e = threadIdx.x; e < TM * BN / 4; e += NT) {
    const int lr = e / (BN / 4), c4 = e - lr * (BN / 4);
    const int64_t row = r0 + lr;
    if (row >= p.n_out) continue;
    float4 v = reinterpret_cast<const float4*>(accT)[e];
    float* dst = p.out + row * (int64_t)cout + n0 + c4 * 4;
    if (p.bias) {
      const float4 bv = *reinterpret_cast<const float4*>(p.bias + n0 + c4 * 4);
      v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
    }
    if (p.accumulate) {
      const float4 o = *reinterpret_cast<const float4*>(dst);
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    *reinterpret_cast<float4*>(dst) = v;
  }
#ifdef USC_PHASE_STATS
  if (threadIdx.x == 0) {
    const long long ph_t4 = clock64();
    USC_PH(0, ph_t1 - ph_t0); USC_PH(1, ph_t3 - ph_t1); USC_PH(2, ph_t4 - ph_t3); USC_PH(6, 1); USC_PH(7, total_items);
  }
#endif
  if (p.stats && threadIdx.x == 0) {
    if (blockIdx.y == 0) {
      int pairs = 0;
      for (int k = 0; k < K; ++k) pairs += cnt[k];
      atomicAdd(p.stats + 2, (unsigned long long)pairs);
    }
    __builtin_amdgcn_s_waitcnt(0);      // (the tile's stores have been issued; the end mark is taken after them)
    atomicMax(p.stats + 1, (unsigned long long)wall_clock64());
  }
}

static size_t compact_lds_bytes(int NB, int TM) {
  return (size_t)(TM + 1) * NB * 32 * 4 + (size_t)kMaxK * TM * 4 + (32 + 32 + 4) * 4 + (size_t)kMaxK * TM + kMaxItems * 4;
}

// out = (accumulate ? out : 0) + bias + sum_g partial[g]   (fixed order)
__global__ __launch_bounds__(256) void group_reduce_kernel(const float* __restrict__ partial, int G, int64_t numel4,
                                                          int cout, const float* __restrict__ bias, int accumulate,
                                                          float* __restrict__ out) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < numel4; j += (int64_t)gridDim.x * blockDim.x) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (accumulate) v = reinterpret_cast<const float4*>(out)[j];
    if (bias) {
      const float4 bv = *reinterpret_cast<const float4*>(bias + (j * 4) % cout);
      v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
    }
    // eight slices in flight per thread (a runtime-G loop waits for every load before the next is issued: 27 dependent
    // round trips for the 27 offset groups of a small map); the adds keep the order g ascending
    int g = 0;
    for (; g + 8 <= G; g += 8) {
      float4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = reinterpret_cast<const float4*>(partial)[(int64_t)(g + u) * numel4 + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) { v.x += t[u].x; v.y += t[u].y; v.z += t[u].z; v.w += t[u].w; }
    }
    for (; g < G; ++g) {
      const float4 t = reinterpret_cast<const float4*>(partial)[(int64_t)g * numel4 + j];
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    reinterpret_cast<float4*>(out)[j] = v;
  }
}

void launch_group_reduce(const float* partial, int G, int64_t numel4, int cout, const float* bias, int accumulate,
                         float* out, hipStream_t st) {   // also used by spconv_sorted.hip
  hipLaunchKernelGGL(group_reduce_kernel, dim3(stream_grid(numel4, 256)), dim3(256), 0, st, partial, G, numel4, cout,
                     bias, accumulate, out);
}

// The stem (reference models/res16unet.py conv0p1s1: 3 colour channels -> 32): with <= 4 input channels a 32-row MFMA
// tile is 90 % zero padding along the reduction (the generic kernel: 107 us, 3.4 TFLOP/s at 150 k voxels).  One thread
// per output row, its 32 output channels in registers; the K*cin weight rows are wave-uniform (scalar loads), the
// neighbour ids of consecutive rows are consecutive in the table.  VALU only: 2*P*cin*32 flop, the launch is bound by the
// 27 dependent id -> row gathers per thread.
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ in, const float* __restrict__ W,
                                                       const int32_t* __restrict__ nbr, const float* __restrict__ bias,
                                                       float* __restrict__ out, int64_t n_out, int K, int cin) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= n_out) return;
  float acc[32];
#pragma unroll
  for (int n = 0; n < 32; ++n) acc[n] = bias ? bias[n] : 0.f;
  for (int k = 0; k < K; ++k) {
    const int v = nbr[(int64_t)k * n_out + row];
    const float* a = in + (int64_t)(v >= 0 ? v : 0) * cin;
    const float keep = v >= 0 ? 1.f : 0.f;
    for (int c = 0; c < cin; ++c) {
      const float av = a[c] * keep;
      const float* w = W + ((int64_t)k * cin + c) * 32;     // uniform address: scalar loads
#pragma unroll
      for (int n = 0; n < 32; ++n) acc[n] = fmaf(av, w[n], acc[n]);
    }
  }
  float4* o = reinterpret_cast<float4*>(out + row * 32);
#pragma unroll
  for (int q = 0; q < 8; ++q) o[q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
}

// Generic path (any cin / cout, e.g. the 3-channel stem and 20-class head): bounds-checked.
template <int NB, bool LIST>
__global__ __launch_bounds__(256) void gather_gemm_kernel(GemmParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.y * (NB * 32);
  const int cin = p.cin, cout = p.cout;
  const bool vec_ok = (cin & 7) == 0;
  int k_begin, k_end;
  int64_t my_out_row, my_in_row_list;
  if (!locate_tile<LIST>(p, wave, i, k_begin, k_end, my_out_row, my_in_row_list)) return;

  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  for (int k = k_begin; k < k_end; ++k) {
    int64_t in_row;
    if (LIST) in_row = my_in_row_list;
    else if (p.nbr) in_row = (my_out_row >= 0) ? (int64_t)p.nbr[(int64_t)k * p.n_out + my_out_row] : -1;
    else in_row = my_out_row;
    const bool valid = in_row >= 0;
    if (!__any(valid)) continue;
    const float* arow = p.in + (valid ? in_row : 0) * (int64_t)cin;
    const float* wk = p.W + (int64_t)k * cin * cout;

    for (int c0 = 0; c0 < cin; c0 += 32) {
      float a[4][4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int c = c0 + 8 * t + 4 * h;
        if (vec_ok) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (valid && c < cin) v = *reinterpret_cast<const float4*>(arow + c);
          a[t][0] = v.x; a[t][1] = v.y; a[t][2] = v.z; a[t][3] = v.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) a[t][j] = (valid && c + j < cin) ? arow[c + j] : 0.f;
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (c0 + 8 * t >= cin) break;  // wave-uniform tail (cin < 32)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = c0 + 8 * t + 4 * h + j;
          const float* wrow = wk + (int64_t)c * cout + n0 + i;
          float b[NB];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) b[nb] = (c < cin && n0 + nb * 32 + i < cout) ? wrow[nb * 32] : 0.f;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = MFMA32(a[t][j], b[nb], acc[nb]);
        }
      }
    }
  }

  float* outp = p.out;
  if (!LIST && p.G > 1) outp += (int64_t)blockIdx.z * p.n_out * cout;
  const bool direct = LIST || p.G == 1;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t orow = __shfl(my_out_row, acc_row(r, h), 64);
    if (orow < 0) continue;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int n = n0 + nb * 32 + i;
      if (n < cout) {
        float v = acc[nb][r];
        float* dst = outp + orow * (int64_t)cout + n;
        if (direct) {
          if (p.bias) v += p.bias[n];
          if (p.accumulate) v += *dst;
        }
        *dst = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// wgrad: dW[k][ci][co] = sum_p a[a_idx[p]][ci] * b[b_idx[p]][co]
struct WgradParams {
  const float* a;
  const float* b;
  const int32_t* a_idx;
  const int32_t* b_idx;
  const int64_t* koff;
  int64_t n_rows;  // identity form
  float* partial;  // [S, K, cin, cout]
  int cin, cout, K, S;
  float* direct;   // wgrad_full_kernel, S == 1: dW itself (no partial slice, no reduction launch)
  int accumulate;  //   ... added into it when set
  int vblocks;     // wgrad_full_kernel: > 0 = number of (problem, offset, split) work items a smaller grid walks
};

// ALIGNED: cin % 32 == 0 and cout % (32*NB) == 0 -> unconditional vector loads; the lane's NB
// columns are contiguous (column of accumulator nb, MFMA column j = co0 + NB*j + nb).
template <int NB, bool ALIGNED>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradParams p) {
  extern __shared__ float red[];  // [3 waves][NB*16 regs][64 lanes]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int k = blockIdx.x / p.S, s = blockIdx.x % p.S;
  const int ci0 = blockIdx.y * 32, co0 = blockIdx.z * (NB * 32);
  const int cin = p.cin, cout = p.cout;

  int64_t pb, pe;
  if (p.a_idx) {
    pb = p.koff[k];
    pe = p.koff[k + 1];
  } else {
    pb = 0;
    pe = p.n_rows;
  }
  // contiguous, even-length chunk of pairs for (split s, wave)
  const int64_t nchunks = (int64_t)p.S * 4;
  int64_t L = (pe - pb + nchunks - 1) / nchunks;
  L = (L + 1) & ~1ll;
  const int64_t cb = pb + ((int64_t)s * 4 + wave) * L;
  int64_t ce = cb + L;
  if (ce > pe) ce = pe;

  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  const bool ci_ok = ci0 + i < cin;
  const int acol = ci0 + i;
  const int bcol = ALIGNED ? co0 + NB * i : co0 + i;
  // 16 pairs per batch: lanes 0-15 fetch a-row ids, 16-31 b-row ids
  auto load_idx = [&](int64_t q) -> int64_t {
    int64_t idxreg = -1;
    const int l16 = lane & 15;
    const int64_t pp = q + l16;
    if (lane < 32 && pp < ce) {
      if (p.a_idx)
        idxreg = (lane < 16) ? p.a_idx[pp] : p.b_idx[pp];
      else
        idxreg = pp;
    }
    return idxreg;
  };
  auto load_rows = [&](int64_t idxreg, float (&av)[8], float (&bv)[8][NB]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t ia = __shfl(idxreg, 2 * u + h, 64);
      const int64_t ib = __shfl(idxreg, 16 + 2 * u + h, 64);
      if (ALIGNED) {
        // out-of-range pairs read row 0; their A value is zeroed, which zeroes the product
        const float t = p.a[(ia >= 0 ? ia : 0) * cin + acol];
        av[u] = ia >= 0 ? t : 0.f;
        load_b<NB>(p.b + (ib >= 0 ? ib : 0) * cout + bcol, bv[u]);
      } else {
        av[u] = (ia >= 0 && ci_ok) ? p.a[ia * cin + acol] : 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          bv[u][nb] = (ib >= 0 && bcol + nb * 32 < cout) ? p.b[ib * cout + bcol + nb * 32] : 0.f;
      }
    }
  };
  auto run = [&](float (&av)[8], float (&bv)[8][NB]) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = MFMA32(av[u], bv[u][nb], acc[nb]);
  };
  if (ALIGNED) {
    // software pipeline: row ids two batches ahead, gathered rows one batch ahead of the MFMAs
    float aX[8], aY[8];
    float bX[8][NB], bY[8][NB];
    int64_t id1 = load_idx(cb), id2 = load_idx(cb + 16);
    load_rows(id1, aX, bX);
    for (int64_t q = cb; q < ce; q += 32) {
      int64_t id3 = load_idx(q + 32);
      load_rows(id2, aY, bY);
      run(aX, bX);
      id2 = load_idx(q + 48);
      load_rows(id3, aX, bX);
      if (q + 16 < ce) run(aY, bY);
    }
  } else {
    float av[8];
    float bv[8][NB];
    for (int64_t q = cb; q < ce; q += 16) {
      load_rows(load_idx(q), av, bv);
      run(av, bv);
    }
  }

  // ordered cross-wave reduction in LDS: wave 0 adds waves 1,2,3 in order
  if (wave > 0) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wave - 1) * NB * 16 + nb * 16 + r) * 64 + lane] = acc[nb][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] += red[(w * NB * 16 + nb * 16 + r) * 64 + lane];
    float* dst = p.partial + ((int64_t)s * p.K + k) * cin * cout;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = ci0 + acc_row(r, h);
      if (ci >= cin) continue;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int co = ALIGNED ? bcol + nb : bcol + nb * 32;
        if (co < cout) dst[(int64_t)ci * cout + co] = acc[nb][r];
      }
    }
  }
}

// wgrad, all of Cin (or half of it) per workgroup: a wave keeps CT x NB accumulator tiles (CT input-channel tiles
// x NB output-column tiles) and walks a contiguous chunk of the offset's pair list; per pair-step it issues TWO
// vector loads (A: a[in][ci0 + CT*i ..+CT), B: b[out][co0 + NB*i ..+NB)) for CT*NB MFMAs.  The older kernel gave every
// 32-channel input tile its own workgroup: each pair's 384-byte b row was gathered Cin/32 times, with 12-byte
// stride-12 loads.  Cross-wave reduction through one LDS tile in three ordered rounds (deterministic).
// GROUPED form (g.R > 0): R problems of ONE shape on ONE kernel map — the weight gradients of a level's residual
// blocks — share the grid: blockIdx.x = (r, k, s).  On the coarse levels a single problem fills a fraction of the chip
// (216 workgroups for 256 -> 256 on 507 rows) and needs a pair split with its reduction launch to reach that; R of
// them in one grid need neither.  Same per-element arithmetic as the single form at the same S.
constexpr int kMaxWgradGroup = 16;
struct WgradGroup { const float* a[kMaxWgradGroup]; const float* b[kMaxWgradGroup]; float* dW[kMaxWgradGroup]; int R; };

// CT * NB <= 9: two workgroups per CU (256 registers per lane).  The 12-tile form <4,3> (128 -> 96 channels, which
// otherwise falls to the per-input-tile kernel) gets the whole register file: 192 accumulator registers in the AGPR
// half, one workgroup per CU.
template <int CT, int NB>
__global__ __launch_bounds__(256, (CT * NB > 9) ? 1 : 2) void wgrad_full_kernel(WgradParams p, WgradGroup g) {
  extern __shared__ float red[];  // [4 waves][NB*16 regs][64 lanes]
  constexpr int NA = CT * NB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  // p.vblocks > 0: a SMALL grid walks the (problem, offset, split) work items — the launch then leaves wave slots and
  // matrix-core time on every CU to whatever runs beside it (the background form of the weight-gradient lane)
  const int vtotal = p.vblocks > 0 ? p.vblocks : (int)gridDim.x;
  for (int vb = blockIdx.x; vb < vtotal; vb += gridDim.x) {
  int bx = vb;
  if (g.R > 0) {
    const int per = p.K * p.S, r = bx / per;
    bx -= r * per;
    p.a = g.a[r]; p.b = g.b[r]; p.direct = g.dW[r];
  }
  const int k = bx / p.S, s = bx % p.S;
  const int ci0 = blockIdx.y * (32 * CT), co0 = blockIdx.z * (NB * 32);
  const int cin = p.cin, cout = p.cout;
  int64_t pb, pe;
  if (p.a_idx) { pb = p.koff[k]; pe = p.koff[k + 1]; } else { pb = 0; pe = p.n_rows; }
  const int64_t nchunks = (int64_t)p.S * 4;
  int64_t L = (pe - pb + nchunks - 1) / nchunks;
  L = (L + 7) & ~7ll;                                  // whole batches of 8 pairs per wave
  const int64_t cb = pb + ((int64_t)s * 4 + wave) * L;
  int64_t ce = cb + L;
  if (ce > pe) ce = pe;

  f32x16 acc[NA];
#pragma unroll
  for (int t = 0; t < NA; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // a batch = 8 pairs = 4 pair-steps (lane half h takes pairs 2u + h); lanes 0-7 fetch a-row ids, 8-15 b-row ids
  auto load_idx = [&](int64_t q) __attribute__((always_inline)) -> int {
    int v = -1;
    const int64_t pp = q + (lane & 7);
    if (lane < 16 && pp < ce) v = p.a_idx ? ((lane < 8) ? p.a_idx[pp] : p.b_idx[pp]) : (int)pp;
    return v;
  };
  auto load_rows = [&](int idxreg, float (&av)[4][CT], float (&bv)[4][NB]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ia = __shfl(idxreg, 2 * u + h, 64);
      const int ib = __shfl(idxreg, 8 + 2 * u + h, 64);
      // MFMA row i of accumulator (ct, .) is input channel ci0 + CT*i + ct, MFMA column i of accumulator (., nb) is
      // output channel co0 + NB*i + nb: a lane's CT A operands and NB B operands are each ONE vector load
      const float* ap = p.a + (int64_t)(ia >= 0 ? ia : 0) * cin + ci0 + CT * i;
      const float* bp = p.b + (int64_t)(ib >= 0 ? ib : 0) * cout + co0 + NB * i;
      const float keep = ia >= 0 ? 1.f : 0.f;          // padding pairs: zero the A operand
      load_b<CT>(ap, av[u]);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) av[u][ct] *= keep;
      load_b<NB>(bp, bv[u]);
    }
  };
  auto run = [&](float (&av)[4][CT], float (&bv)[4][NB]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[ct * NB + nb] = MFMA32(av[u][ct], bv[u][nb], acc[ct * NB + nb]);
  };
  float aX[4][CT], aY[4][CT], bX[4][NB], bY[4][NB];
  int id1 = load_idx(cb), id2 = load_idx(cb + 8);
  load_rows(id1, aX, bX);
#ifdef USC_ABLATE_WG_LOOP
  if (p.cin < 0)
#endif
  for (int64_t q = cb; q < ce; q += 16) {
    const int id3 = load_idx(q + 16);
    load_rows(id2, aY, bY);
    run(aX, bX);
    id2 = load_idx(q + 24);
    load_rows(id3, aX, bX);
    if (q + 8 < ce) run(aY, bY);
  }
  // (three row buffers with the ids four batches ahead measured 7 % SLOWER: the loop is bound by the matrix-core pipe that
  //  two resident workgroups share — 1.8 GFLOP = 11 us at peak for the 507-row 256->256 layer, 15 us measured — not by
  //  its loads)

#ifdef USC_ABLATE_WG_EPI
  if (p.cin > 0) {
    if (wave == 0 && lane == 0) p.partial[((int64_t)s * p.K + k) * cin * cout + ci0 * cout + co0] = acc[0][0] + acc[NA - 1][15];
    return;
  }
#endif
  // Reduction over the four waves, in the fixed order w0 + ((w3 + w2) + w1) per element, one input-channel tile (ct)
  // at a time: every wave parks the tile's NB*16 accumulators in LDS, then wave w sums accumulator rows 4w .. 4w+3
  // over the four waves and writes them — all four waves add and store (the first version let wave 3, 2, 1 add in
  // turn and wave 0 write everything with 4-byte stores: 7 us of a 35 us launch on the 507-row level), and a lane's NB
  // columns of one row are adjacent in dW: one 16-byte store for NB = 4.
  float* dst = (p.direct ? p.direct : p.partial + (int64_t)s * p.K * cin * cout) + (int64_t)k * cin * cout;
  const bool rmw = p.direct && p.accumulate;   // one split: added straight into dW (no partial slice, no reduction launch)
  const bool vec_ok = (reinterpret_cast<uintptr_t>(dst) & (NB * 4 - 1)) == 0;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    if (ct) __syncthreads();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wave * NB + nb) * 16 + r) * 64 + lane] = acc[ct * NB + nb][r];
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = 4 * wave + rr;
      float v[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float t = red[((3 * NB + nb) * 16 + r) * 64 + lane];
        t += red[((2 * NB + nb) * 16 + r) * 64 + lane];
        t += red[((1 * NB + nb) * 16 + r) * 64 + lane];
        v[nb] = red[(nb * 16 + r) * 64 + lane] + t;
      }
      const int ci = ci0 + CT * acc_row(r, h) + ct;
      float* o = dst + (int64_t)ci * cout + co0 + NB * i;
      if (NB == 4 && vec_ok) {
        float4 w4 = make_float4(v[0], v[1], v[2 % NB], v[3 % NB]);
        if (rmw) {
          const float4 old = *reinterpret_cast<const float4*>(o);
          w4.x += old.x; w4.y += old.y; w4.z += old.z; w4.w += old.w;
        }
        *reinterpret_cast<float4*>(o) = w4;
      } else if (NB == 2 && vec_ok) {
        float2 w2 = make_float2(v[0], v[1 % NB]);
        if (rmw) {
          const float2 old = *reinterpret_cast<const float2*>(o);
          w2.x += old.x; w2.y += old.y;
        }
        *reinterpret_cast<float2*>(o) = w2;
      } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) o[nb] = rmw ? o[nb] + v[nb] : v[nb];
      }
    }
  }
  __syncthreads();   // `red` is reused by the next work item
  }
}

// Weight gradient of the stem (<= 4 input channels, 32 output channels), table form: dW[k][c][n] = sum_o in[nbr[k][o]][c] dy[o][n].
// The pair-list kernels run one offset per workgroup: a [32 x 32] MFMA tile whose M side holds 3 useful rows (131 us,
// 2.9 TFLOP/s at 150 k voxels).  Here the M side holds 32 / cin offsets x cin channels — MFMA row m = (offset j, channel c)
// — so ceil(K / (32 / cin)) = 3 groups replace 27 passes over the rows; dW row k*cin + c = group*per*cin + m is
// contiguous, the groups' tiles are slices of dW itself.  Lane (i, h) feeds A[m = i][row 2t + h] = its own (offset,
// channel) element of that row's neighbour and B[row 2t + h][n = i] = dy; S row ranges -> partial slices in dW layout,
// summed in order by wgrad_reduce_kernel.
constexpr int kStemSplits = 64;
constexpr int kStemWaves = 16;      // a trip is two dependent loads (id -> element) and four MFMAs: latency hidden by waves
__global__ __launch_bounds__(64 * kStemWaves) void stem_wgrad_kernel(const float* __restrict__ in, const float* __restrict__ dy,
                                                                    const int32_t* __restrict__ nbr, int64_t n_out, int K,
                                                                    int cin, float* __restrict__ partial) {
  __shared__ float red[kStemWaves][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
  const int per = 32 / cin;                                  // offsets per group
  const int g = blockIdx.y, s = blockIdx.x, S = gridDim.x;
  const int j = i / cin, c = i - j * cin, k = g * per + j;
  const bool active = j < per && k < K;
  int64_t L = (n_out + S - 1) / S;
  L = (L + 2 * kStemWaves - 1) / (2 * kStemWaves) * (2 * kStemWaves);
  const int64_t r_begin = (int64_t)s * L;
  int64_t r_end = r_begin + L;
  if (r_end > n_out) r_end = n_out;
  const int32_t* nk = nbr + (int64_t)(active ? k : 0) * n_out;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // four rounds in flight: ids, then the gathered elements and dy, then the matrix cores
  for (int64_t rb = r_begin + 2 * wave; rb < r_end; rb += 8 * kStemWaves) {   // (wave-uniform trip count: MFMA needs every lane)
    int id[4];
    float b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t r = rb + h + 2 * kStemWaves * u;
      const bool ok = r < r_end;
      id[u] = (ok && active) ? nk[r] : -1;
      b[u] = ok ? dy[r * 32 + i] : 0.f;
    }
    float a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = id[u] >= 0 ? in[(int64_t)id[u] * cin + c] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  if (wave >= 4) return;
  float* dst = partial + (int64_t)s * K * cin * 32;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 4 * wave + q;
    const int m = acc_row(r, h);
    float v = red[kStemWaves - 1][r][lane];
#pragma unroll
    for (int w = kStemWaves - 2; w >= 0; --w) v += red[w][r][lane];
    const int row = g * per * cin + m;
    if (m < per * cin && row < K * cin) dst[(int64_t)row * 32 + i] = v;
  }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, int S, int64_t numel, int accumulate,
                                    float* __restrict__ dW) {
  // float4 per thread when the slice length allows it (every [K, cin, cout] with cout % 4 == 0), slices in s order
  if ((numel & 3) == 0 && (reinterpret_cast<uintptr_t>(dW) & 15) == 0) {   // (a .grad view may sit at any float offset)
    const int64_t n4 = numel >> 2;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n4; j += (int64_t)gridDim.x * blockDim.x) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      // eight slices in flight per thread (one load per iteration was one memory round trip per slice: 51 dependent trips
      // for the 12 800-row projections of the decoder, 64 on the finest level); the adds keep the slice order
      int s = 0;
      for (; s + 8 <= S; s += 8) {
        float4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = reinterpret_cast<const float4*>(partial)[(int64_t)(s + u) * n4 + j];
#pragma unroll
        for (int u = 0; u < 8; ++u) { v.x += t[u].x; v.y += t[u].y; v.z += t[u].z; v.w += t[u].w; }
      }
      for (; s < S; ++s) {
        const float4 t = reinterpret_cast<const float4*>(partial)[(int64_t)s * n4 + j];
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      if (accumulate) {      // straight into the parameter's gradient buffer
        const float4 o = reinterpret_cast<const float4*>(dW)[j];
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
      }
      reinterpret_cast<float4*>(dW)[j] = v;
    }
    return;
  }
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < numel; j += (int64_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += partial[(int64_t)s * numel + j];
    dW[j] = accumulate ? dW[j] + v : v;   // accumulate: straight into the parameter's gradient buffer
  }
}

// out[k][co][ci] = W[mirror ? K-1-k : k][ci][co]
__global__ void weight_transpose_kernel(const float* __restrict__ W, int K, int cin, int cout, int mirror,
                                        float* __restrict__ out) {
  const int64_t per = (int64_t)cin * cout;
  const int64_t total = per * K;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(j / per);
    const int64_t rem = j - (int64_t)k * per;
    const int co = (int)(rem / cin), ci = (int)(rem - (int64_t)co * cin);
    const int ks = mirror ? (K - 1 - k) : k;
    out[j] = W[(int64_t)ks * per + (int64_t)ci * cout + co];
  }
}

// Weight slices in matrix-core operand order for the tile-compacted kernel:
//   Wp[cb][k][q][nb][lane = 32h + i][j] = W[k][8q + 4h + j][cb*32*NB + NB*i + nb]
// so that the four B operands a lane needs for one quad and one accumulator are ONE 16-byte load and the
// wave's 64 loads are 1 KiB contiguous.  (The natural [cin][cout] layout made every k-step a 12-byte-per-lane,
// stride-12 load that the texture path splits into strided dword passes: measured 85 cycles per wave
// instruction and 0.55 ms of a 0.60 ms kernel bound by it, independent of prefetch depth.)
// wt = 1: W is the forward conv's [K, cout, cin] and the packed slices are those of W'[k][c][n] = W[K-1-k][n][c]
// (input gradient of a stride-1 conv) — the transpose pass is folded into the packing.
__global__ void weight_pack_kernel(const float* __restrict__ W, int K, int cin, int cout, int NB, int wt,
                                   float* __restrict__ out) {
  const int64_t total = (int64_t)K * cin * cout;
  const int nq = cin >> 3, BN = NB * 32;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = e;
    const int j = (int)(r & 3); r >>= 2;
    const int ln = (int)(r & 63); r >>= 6;
    const int nb = (int)(r % NB); r /= NB;
    const int q = (int)(r % nq); r /= nq;
    const int k = (int)(r % K);
    const int cb = (int)(r / K);
    const int i = ln & 31, h = ln >> 5;
    const int c = 8 * q + 4 * h + j, n = cb * BN + NB * i + nb;
    out[e] = wt ? W[((int64_t)(K - 1 - k) * cout + n) * cin + c] : W[((int64_t)k * cin + c) * cout + n];
  }
}

static int wgrad_splits(int K, int cin, int cout, int NB, int64_t n_rows) {
  const int64_t tiles = (int64_t)K * ceil_div(cin, 32) * ceil_div(cout, NB * 32);
  // measured on the 128->96 layers (tools/scratch/r03/wg_target2.sh): 4.0 M pairs 902 / 776 / 730 / 745 us at 1 024 /
  // 2 048 / 4 096 / 6 912 workgroups, 1.1 M pairs 286 / 266 / 276 / 303 us
  static const int knob = getenv("USC3D_WGRAD_TILE_TARGET") ? atoi(getenv("USC3D_WGRAD_TILE_TARGET")) : 0;
  const int target = knob > 0 ? knob : ((K > 1 && n_rows >= (int64_t)2 << 20) ? 4096 : 2048);
  int64_t S = ceil_div(target, tiles);
  const int64_t by_rows = n_rows / 2048 + 1;
  if (S > by_rows) S = by_rows;
  if (S > 64) S = 64;
  if (S < 1) S = 1;
  return (int)S;
}
static int wgrad_full_nb(int cb) { return (cb % 3 == 0) ? 3 : (cb % 4 == 0 ? 4 : (cb % 2 == 0 ? 2 : 1)); }
static bool wgrad_big_tiles() {      // USC3D_WGRAD_BIG=0: without the 12-tile forms (A/B switch)
  static const bool on = !getenv("USC3D_WGRAD_BIG") || atoi(getenv("USC3D_WGRAD_BIG")) != 0;
  return on;
}
static int wgrad_full_ct(int ctiles, int NBf) {
  if (wgrad_big_tiles()) {
    if (NBf == 3 && ctiles % 3 != 0 && ctiles % 4 == 0) return 4;     // e.g. 128 -> 96: 148 -> 134 us per launch on average
    // (the mirror form <3,4> for 96 -> 128 measured SLOWER than one input tile per workgroup, 167 vs 75 us on the K = 1
    //  layer of the finest level: not used)
  }
  return (ctiles % 3 == 0 && NBf * 3 <= 9) ? 3 : ((ctiles % 4 == 0 && NBf * 4 <= 9) ? 4 : ((ctiles % 2 == 0 && NBf * 2 <= 9) ? 2 : 1));
}
static int64_t wgrad_full_splits(int K, int ctiles, int cb, int CT, int NBf, int64_t n_rows) {
  const int64_t blocks_per_split = (int64_t)K * (ctiles / CT) * (cb / NBf);
  static const int target = getenv("USC3D_WGRAD_TARGET_BLOCKS") ? atoi(getenv("USC3D_WGRAD_TARGET_BLOCKS")) : 512;
  static const int rows_per = getenv("USC3D_WGRAD_ROWS_PER_SPLIT") ? atoi(getenv("USC3D_WGRAD_ROWS_PER_SPLIT")) : 4096;
  // 512: one round of 2 workgroups per CU.  The finest level (4.0 M pairs at 150 k voxels) is the exception: 64 slices
  // of its 27 x 1 x 1 workgroups (1 728, ~0.9 MB each) measured 417 us against 451 us with 18 (86.5 vs 80 TFLOP/s) — the
  // long pair lists leave a tail of half-empty CUs that more, shorter workgroups fill; from 1.1 M pairs down the extra
  // slices cost more than they return (158 -> 164 us, 33 -> 38 us on the coarse levels).
  const int64_t tgt = (K > 1 && n_rows >= (int64_t)2 << 20 && target < 2048) ? 2048 : target;
  int64_t S = tgt / blocks_per_split;
  // USC3D_WGRAD_ONE_SLICE_FROM=<blocks>: no pair split (and no reduction launch) once a single slice already has that
  // many workgroups (experiment knob)
  static const int one_from = getenv("USC3D_WGRAD_ONE_SLICE_FROM") ? atoi(getenv("USC3D_WGRAD_ONE_SLICE_FROM")) : 0;
  if (one_from > 0 && blocks_per_split >= one_from) S = 1;
  const int64_t by_rows = n_rows / (K > 1 ? rows_per : 256) + 1;   // identity pairs (dense layers): 64 rows per wave suffice
  if (S > by_rows) S = by_rows;
  if (S > 64) S = 64;
  if (S < 1) S = 1;
  return S;
}
static int pick_nb(int cout) {
  if (cout <= 32) return 1;
  if (cout <= 64) return 2;
  if (cout % 128 == 0) return 4;
  if (cout % 96 == 0) return 3;
  return 4;
}

// Launch plan of the table-form gather GEMM: column blocks per wave (NB) and offset
// groups (G).  Large maps keep the widest tile (least re-gathering of A rows); small maps
// (coarse U-Net levels: hundreds to a few thousand rows, 128-256 channels) shrink the tile
// and split the K offsets over blockIdx.z so that >= ~3 waves per SIMD are in flight.
struct GemmPlan { int NB; int G; bool aligned; int TM; };   // TM > 0: tile-compacted kernel
constexpr int64_t kTargetWaves = 3072;

// few input channels, 32 output channels, a neighbour table: stem_conv_kernel
static bool stem_form(bool has_table, int cin, int cout, int K) {
  static const bool on = !getenv("USC3D_STEM_KERNEL") || atoi(getenv("USC3D_STEM_KERNEL")) != 0;
  return on && has_table && cin >= 1 && cin <= 4 && cout == 32 && K >= 1 && K <= 32;
}

static GemmPlan plan_table(int64_t n_out, int cin, int cout, int K) {
  GemmPlan pl{pick_nb(cout), 1, false, 0};
  const bool al = (cin % 32 == 0) && (cout % 32 == 0);
  if (!al) return pl;
  pl.aligned = true;
  if (K > 1 && K <= kMaxK && n_out >= 24576 && cin >= 64 && cin <= kZeroRowFloats) {
    // large maps: tile-compacted kernel (no MFMA work on absent neighbours).  One 8-wave workgroup
    // per CU; the tile height is chosen so that the tiles fill an integer number of 256-CU rounds.
    const int cb = cout / 32;
    const int nb = (cb % 3 == 0) ? 3 : (cb % 2 == 0 ? 2 : 1);
#ifndef USC_TM_MAX3
#define USC_TM_MAX3 192
#endif
    const int tm_max = nb == 3 ? USC_TM_MAX3 : 256;
#ifndef USC_TILE_SLOTS
#define USC_TILE_SLOTS 256   /* whole rounds of 256 CUs; 512 (both resident workgroups of a CU) measured 5 % slower on the 40 k-row map: smaller tiles pad more */
#endif
    int64_t R = 1;
    int64_t tm = ceil_div(n_out, (int64_t)USC_TILE_SLOTS * R);
    while (tm > tm_max) { ++R; tm = ceil_div(n_out, (int64_t)USC_TILE_SLOTS * R); }
    tm = (tm + 3) & ~3ll;
    pl.NB = nb;
    pl.TM = (int)tm;
    return pl;
  }
  const int64_t row_tiles = ceil_div(n_out, 32);
  const int cb = cout / 32;
  const int cands[4] = {4, 3, 2, 1};
  int chosen = 1;
  for (int c = 0; c < 4; ++c) {
    const int nb = cands[c];
    if (cb % nb) continue;
    if (row_tiles * (cb / nb) >= kTargetWaves) { chosen = nb; break; }
  }
  pl.NB = chosen;
  const int64_t waves = row_tiles * (cb / chosen);
  if (waves > 0 && waves < kTargetWaves && K > 1) {   // (an empty map plans to nothing)
    int64_t G = ceil_div(kTargetWaves, waves);
    if (G > K) G = K;
    pl.G = (int)G;
  }
  return pl;
}
static GemmPlan plan_list(int64_t max_tiles, int cin, int cout) {
  GemmPlan pl{pick_nb(cout), 1, false, 0};
  const bool al = (cin % 32 == 0) && (cout % 32 == 0);
  if (!al) return pl;
  pl.aligned = true;
  const int cb = cout / 32;
  const int cands[4] = {4, 3, 2, 1};
  int chosen = 1;
  for (int c = 0; c < 4; ++c) {
    const int nb = cands[c];
    if (cb % nb) continue;
    if (max_tiles * (cb / nb) >= kTargetWaves) { chosen = nb; break; }
  }
  pl.NB = chosen;
  return pl;
}

template <bool LIST>
static void launch_gemm(const GemmPlan& pl, dim3 grid, hipStream_t st, const GemmParams& p) {
#define USC_GG(NBv)                                                                                      \
  if (pl.aligned && p.wt1)       /* compile-time variant: a runtime test in the loop cost the plain form 35 % */ \
    hipLaunchKernelGGL((gather_gemm_aligned_kernel<NBv, false, true>), grid, dim3(256), 0, st, p);      \
  else if (pl.aligned)                                                                                   \
    hipLaunchKernelGGL((gather_gemm_aligned_kernel<NBv, LIST>), grid, dim3(256), 0, st, p);             \
  else                                                                                                   \
    hipLaunchKernelGGL((gather_gemm_kernel<NBv, LIST>), grid, dim3(256), 0, st, p);
  switch (pl.NB) {
    case 1: USC_GG(1) break;
    case 2: USC_GG(2) break;
    case 3: USC_GG(3) break;
    default: USC_GG(4) break;
  }
#undef USC_GG
}

}  // namespace usc

using namespace usc;

extern "C" {

int usc_weight_transpose(const float* W, int32_t K, int32_t cin, int32_t cout, int32_t mirror, float* out,
                         usc_stream_t s) {
  USC_REQUIRE(W && out && K >= 1 && cin >= 1 && cout >= 1, "usc_weight_transpose: bad argument");
  const int64_t total = (int64_t)K * cin * cout;
  hipLaunchKernelGGL(weight_transpose_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(s), W, (int)K,
                     (int)cin, (int)cout, (int)mirror, out);
  USC_CHECK_LAUNCH("usc_weight_transpose");
  return USC_OK;
}

int usc_spconv_plan(int32_t kind, int64_t n, int32_t cin, int32_t cout, int32_t K) {
  // kind 0: gather_gemm on n output rows; 1: pairs_gemm with P_capacity n; 2: wgrad
  GemmPlan pl;
  if (kind == 0 && stem_form(K > 1, cin, cout, K)) return 1 | (1 << 14);
  if (kind == 0) pl = plan_table(n, cin, cout, K);
  else if (kind == 1) pl = plan_list(ceil_div(n, 32) + K, cin, cout);
  else {
    // wgrad: mirror usc_spconv_wgrad's choice (full kernel: NB | 1<<8 | 1<<13 | CT<<16)
    const int ctiles = cin / 32, cb = cout / 32;
    const int NBf = (cb % 3 == 0) ? 3 : (cb % 4 == 0 ? 4 : (cb % 2 == 0 ? 2 : 1));
    if (cin % 32 == 0 && cout % 32 == 0 && (!(NBf == 3 && ctiles % 3 != 0) || (wgrad_big_tiles() && ctiles % 4 == 0))) {
      const int CT = wgrad_full_ct(ctiles, NBf);
      return NBf | (1 << 8) | (1 << 13) | (CT << 16);
    }
    const int NB = pick_nb(cout);
    pl = GemmPlan{NB, 1, (cin % 32 == 0) && (cout % (NB * 32) == 0), 0};
  }
  return pl.NB | ((pl.aligned ? 1 : 0) << 8) | ((pl.TM > 0 ? 1 : 0) << 12) | (pl.G << 16);
}

int64_t usc_spconv_gather_gemm_ws_bytes(int64_t n_out, int32_t cin, int32_t cout, int32_t K) {
  const GemmPlan pl = plan_table(n_out, cin, cout, K);
  if (pl.TM > 0) return (int64_t)K * cin * cout * 4;   // packed weights (weight_pack_kernel)
  return pl.G > 1 ? (int64_t)pl.G * n_out * cout * 4 : 0;
}

// ---- launch statistics of the tile-compacted kernel, taken by the kernel itself inside whatever step is running ----
namespace {
struct LaunchStats { unsigned long long* ring = nullptr; int64_t slots = 0, count = 0; std::vector<usc_launch_stat> meta; };
LaunchStats g_lstats;
__global__ void launch_stats_init_kernel(unsigned long long* ring, int64_t slots) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < slots) { ring[4 * i] = ~0ull; ring[4 * i + 1] = 0; ring[4 * i + 2] = 0; ring[4 * i + 3] = 0; }
}
}  // namespace

int usc_launch_stats_begin(void* ring_dev, int64_t slots, usc_stream_t s) {
  USC_REQUIRE(ring_dev && slots > 0, "usc_launch_stats_begin: needs a device ring of slots x 4 x u64");
  hipLaunchKernelGGL(launch_stats_init_kernel, dim3((unsigned)ceil_div(slots, 256)), dim3(256), 0, as_stream(s),
                     (unsigned long long*)ring_dev, slots);
  USC_CHECK_LAUNCH("usc_launch_stats_begin");
  g_lstats.ring = (unsigned long long*)ring_dev;
  g_lstats.slots = slots;
  g_lstats.count = 0;
  g_lstats.meta.clear();
  g_lstats.meta.reserve((size_t)slots);
  return USC_OK;
}

int64_t usc_launch_stats_end(usc_launch_stat* host_out, int64_t max_out) {
  const int64_t n = g_lstats.count;
  if (host_out)
    for (int64_t i = 0; i < n && i < max_out; ++i) host_out[i] = g_lstats.meta[(size_t)i];
  g_lstats.ring = nullptr;
  g_lstats.slots = g_lstats.count = 0;
  g_lstats.meta.clear();
  return n;
}

#ifdef USC_PHASE_STATS
// developer build only: read (and clear) the phase counters of gather_gemm_compact_kernel
extern "C" int usc_phase_stats_read(unsigned long long* out16) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 16) != hipSuccess) return USC_ERR_LAUNCH;
  unsigned long long z[16] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) == hipSuccess ? USC_OK : USC_ERR_LAUNCH;
}
#endif

int64_t usc_wall_clock_khz(void) {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
  return khz;
}

int usc_spconv_gather_gemm(const float* in, int64_t n_in, int32_t cin, const float* W, int32_t K, int32_t cout,
                           const int32_t* nbr, int64_t n_out, const float* bias, float* out, int32_t accumulate,
                           int32_t w_transposed, void* ws, int64_t ws_bytes, usc_stream_t s) {
  USC_REQUIRE(cin >= 1 && cout >= 1 && K >= 1 && n_out >= 0 && n_in >= 0, "usc_spconv_gather_gemm: bad sizes");
  USC_REQUIRE(nbr || K == 1, "usc_spconv_gather_gemm: K>1 needs a neighbour table");
  USC_REQUIRE(nbr || n_in == n_out, "usc_spconv_gather_gemm: identity map needs n_in == n_out");
  if (n_out == 0) return USC_OK;
  USC_REQUIRE(in && W && out, "usc_spconv_gather_gemm: null pointer");
  if (stem_form(nbr != nullptr, cin, cout, K) && !accumulate && !w_transposed) {
    hipLaunchKernelGGL(stem_conv_kernel, dim3((unsigned)ceil_div(n_out, 256)), dim3(256), 0, as_stream(s), in, W, nbr, bias,
                       out, n_out, (int)K, (int)cin);
    USC_CHECK_LAUNCH("usc_spconv_gather_gemm");
    return USC_OK;
  }
  const GemmPlan pl = plan_table(n_out, cin, cout, K);
  GemmParams p{};
  p.in = in; p.W = W; p.nbr = nbr; p.bias = bias; p.out = out;
  p.n_out = n_out; p.cin = cin; p.cout = cout; p.K = K; p.accumulate = accumulate; p.G = pl.G;
  if (pl.G > 1) {
    USC_REQUIRE(ws && ws_bytes >= (int64_t)pl.G * n_out * cout * 4, "usc_spconv_gather_gemm: workspace too small");
    p.out = (float*)ws;
  }
  hipStream_t st = as_stream(s);
  const bool wt1 = w_transposed && K == 1 && !nbr && pl.aligned && pl.TM == 0;   // a linear layer's [cout][cin] weight
  USC_REQUIRE(!w_transposed || pl.TM > 0 || wt1,
              "usc_spconv_gather_gemm: w_transposed is only folded into the tile-compacted kernel (usc_spconv_plan bit 12) "
              "and, for K = 1 on identity rows with channel counts in multiples of 32, into the row-order kernel; "
              "use usc_weight_transpose for other shapes");
  p.wt1 = wt1 ? 1 : 0;
  if (pl.TM > 0) {
    p.TM = pl.TM;
    USC_REQUIRE(ws && ws_bytes >= (int64_t)K * cin * cout * 4, "usc_spconv_gather_gemm: workspace too small");
    hipLaunchKernelGGL(weight_pack_kernel, dim3(stream_grid((int64_t)K * cin * cout, 256)), dim3(256), 0, st, W, (int)K,
                       (int)cin, (int)cout, pl.NB, (int)w_transposed, (float*)ws);
    p.W = (const float*)ws;
    dim3 cgrid((unsigned)ceil_div(n_out, pl.TM), (unsigned)(cout / (pl.NB * 32)));
    const size_t lds = compact_lds_bytes(pl.NB, pl.TM);
    if (g_lstats.ring && g_lstats.count < g_lstats.slots) {
      p.stats = g_lstats.ring + 4 * g_lstats.count++;
      g_lstats.meta.push_back(usc_launch_stat{n_out, cin, cout, K, pl.NB});
    }
#define USC_CG(NBv)                                                                                        \
    {                                                                                                      \
      static bool attr_set = false;                                                                        \
      auto kfn = gather_gemm_compact_kernel<NBv>;                                                          \
      if (!attr_set) {                                                                                     \
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        attr_set = true;                                                                                   \
      }                                                                                                    \
      hipLaunchKernelGGL(kfn, cgrid, dim3(64 * kCompactWaves), lds, st, p);                                \
    }
    if (pl.NB == 3) USC_CG(3) else if (pl.NB == 2) USC_CG(2) else USC_CG(1)
#undef USC_CG
    USC_CHECK_LAUNCH("usc_spconv_gather_gemm");
    return USC_OK;
  }
  dim3 grid((unsigned)ceil_div(n_out, 128), (unsigned)ceil_div(cout, pl.NB * 32), (unsigned)pl.G);
  launch_gemm<false>(pl, grid, st, p);
  if (pl.G > 1) {
    const int64_t numel4 = n_out * cout / 4;
    hipLaunchKernelGGL(group_reduce_kernel, dim3(stream_grid(numel4, 256)), dim3(256), 0, st, (const float*)ws, pl.G,
                       numel4, (int)cout, bias, (int)accumulate, out);
  }
  USC_CHECK_LAUNCH("usc_spconv_gather_gemm");
  return USC_OK;
}

int usc_spconv_pairs_gemm(const float* in, int32_t cin, const float* W, int32_t K, int32_t cout,
                          const int32_t* rows_in, const int32_t* rows_out, const int64_t* koff, int64_t P_capacity,
                          float* out, usc_stream_t s) {
  USC_REQUIRE(cin >= 1 && cout >= 1 && K >= 1 && P_capacity >= 0, "usc_spconv_pairs_gemm: bad sizes");
  if (P_capacity == 0) return USC_OK;
  USC_REQUIRE(in && W && rows_in && rows_out && koff && out, "usc_spconv_pairs_gemm: null pointer");
  GemmParams p{};
  p.in = in; p.W = W; p.out = out; p.cin = cin; p.cout = cout; p.K = K; p.G = 1;
  p.rows_in = rows_in; p.rows_out = rows_out; p.koff = koff;
  const int64_t max_tiles = ceil_div(P_capacity, 32) + K;  // each offset pads to a multiple of 32 pairs
  const GemmPlan pl = plan_list(max_tiles, cin, cout);
  dim3 grid((unsigned)ceil_div(max_tiles, 4), (unsigned)ceil_div(cout, pl.NB * 32));
  launch_gemm<true>(pl, grid, as_stream(s), p);
  USC_CHECK_LAUNCH("usc_spconv_pairs_gemm");
  return USC_OK;
}

int64_t usc_spconv_wgrad_ws_bytes(int32_t K, int32_t cin, int32_t cout) {
  return (int64_t)64 * K * cin * cout * (int64_t)sizeof(float);
}

int64_t usc_spconv_wgrad_ws_bytes_rows(int32_t K, int32_t cin, int32_t cout, int64_t n_rows) {
  // the split count usc_spconv_wgrad will choose for this pair-list capacity (same arithmetic)
  const int64_t numel = (int64_t)K * cin * cout;
#ifndef USC_WGRAD_LEGACY
  const int ctiles = cin / 32, cb = cout / 32;
  if (cin % 32 == 0 && cout % 32 == 0 && cb > 0) {
    const int NBf = wgrad_full_nb(cb);
    if (!(NBf == 3 && ctiles % 3 != 0) || (wgrad_big_tiles() && ctiles % 4 == 0))
      return wgrad_full_splits(K, ctiles, cb, wgrad_full_ct(ctiles, NBf), NBf, n_rows) * numel * 4;
  }
#endif
  return (int64_t)wgrad_splits(K, cin, cout, pick_nb(cout), n_rows) * numel * 4;
}

int64_t usc_spconv_wgrad_table_ws_bytes(int32_t K, int32_t cin, int32_t cout) {
  return (int64_t)kStemSplits * K * cin * cout * 4;
}

int usc_spconv_wgrad_table(const float* in, int32_t cin, const float* dy, int32_t cout, const int32_t* nbr, int32_t K,
                           int64_t n_out, float* dW, int32_t accumulate, void* ws, int64_t ws_bytes, usc_stream_t s) {
  USC_REQUIRE(cin >= 1 && cin <= 4 && cout == 32 && K >= 1 && K <= 32 && n_out >= 0,
              "usc_spconv_wgrad_table: the table form covers <= 4 input channels and 32 output channels (the stem)");
  USC_REQUIRE(in && dy && nbr && dW && ws, "usc_spconv_wgrad_table: null pointer");
  USC_REQUIRE(ws_bytes >= usc_spconv_wgrad_table_ws_bytes(K, cin, cout), "usc_spconv_wgrad_table: workspace too small");
  hipStream_t st = as_stream(s);
  const int per = 32 / cin, groups = (K + per - 1) / per;
  const int64_t numel = (int64_t)K * cin * 32;
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3(kStemSplits, (unsigned)groups), dim3(64 * kStemWaves), 0, st, in, dy, nbr, n_out, (int)K,
                     (int)cin, (float*)ws);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(stream_grid(numel, 256)), dim3(256), 0, st, (const float*)ws, kStemSplits, numel,
                     (int)accumulate, dW);
  USC_CHECK_LAUNCH("usc_spconv_wgrad_table");
  return USC_OK;
}

// > 0: usc_spconv_wgrad launches its all-input-tiles kernel with at most this many workgroups along x, each walking
// several (offset, split) work items — set around the launches of the weight-gradient lane (units.hip), which run
// BESIDE the backward pass's own chain and must leave it wave slots on every CU
static int g_wgrad_grid_limit = getenv("USC3D_WGRAD_GRID_LIMIT") ? atoi(getenv("USC3D_WGRAD_GRID_LIMIT")) : 0;
void usc_spconv_wgrad_grid_limit(int32_t max_workgroups) { g_wgrad_grid_limit = max_workgroups; }

int usc_spconv_wgrad(const float* a, int32_t cin, const float* b, int32_t cout, int32_t K, const int32_t* a_idx,
                     const int32_t* b_idx, const int64_t* koff, int64_t n_rows, float* dW, int32_t accumulate, void* ws, int64_t ws_bytes,
                     usc_stream_t s) {
  USC_REQUIRE(cin >= 1 && cout >= 1 && K >= 1 && n_rows >= 0, "usc_spconv_wgrad: bad sizes");
  USC_REQUIRE(a && b && dW && ws, "usc_spconv_wgrad: null pointer");
  USC_REQUIRE((a_idx && b_idx && koff) || (!a_idx && K == 1), "usc_spconv_wgrad: pair lists required for K>1");
  const int64_t numel = (int64_t)K * cin * cout;
  hipStream_t st = as_stream(s);
  WgradParams p{};
  p.a = a; p.b = b; p.a_idx = a_idx; p.b_idx = b_idx; p.koff = koff; p.n_rows = n_rows;
  p.partial = (float*)ws; p.cin = cin; p.cout = cout; p.K = K;
#ifndef USC_WGRAD_LEGACY
  const int ctiles = cin / 32, cb = cout / 32;
  const int NBf = wgrad_full_nb(cb);
  // (128 -> 96 channels would need 4x3 tiles = 256 VGPRs + scratch, or 2x3 twice: both measured slower than the
  //  per-input-tile kernel below, 0.78 / 0.82 vs 0.73 ms)
  if (cin % 32 == 0 && cout % 32 == 0 && (!(NBf == 3 && ctiles % 3 != 0) || (wgrad_big_tiles() && ctiles % 4 == 0))) {
    // CT input tiles x NB column tiles per wave, CT*NB <= 9 accumulator tiles
    const int CT = wgrad_full_ct(ctiles, NBf);
    const int64_t S = wgrad_full_splits(K, ctiles, cb, CT, NBf, n_rows);
    USC_REQUIRE(ws_bytes >= S * numel * 4, "usc_spconv_wgrad: workspace too small");
    p.S = (int)S;
    if (S == 1) { p.direct = dW; p.accumulate = accumulate; }   // one slice: written (added) straight into dW
    dim3 grid((unsigned)(K * S), (unsigned)(ctiles / CT), (unsigned)(cb / NBf));
    if (g_wgrad_grid_limit > 0 && (int64_t)K * S > g_wgrad_grid_limit) {   // background form (usc_spconv_wgrad_grid_limit)
      p.vblocks = (int)(K * S);
      grid.x = (unsigned)g_wgrad_grid_limit;
    }
    const size_t lds = (size_t)4 * NBf * 16 * 64 * sizeof(float);   // four waves x one input-channel tile
#define USC_WF(C, N) if (CT == C && NBf == N) { \
      static bool attr_set = false; auto kfn = wgrad_full_kernel<C, N>; \
      if (!attr_set) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); attr_set = true; } \
      hipLaunchKernelGGL(kfn, grid, dim3(256), lds, st, p, WgradGroup{}); }
    USC_WF(3, 3) USC_WF(2, 3) USC_WF(1, 3) USC_WF(2, 4) USC_WF(1, 4) USC_WF(4, 2) USC_WF(3, 2) USC_WF(2, 2) USC_WF(1, 2) USC_WF(4, 1) USC_WF(3, 1) USC_WF(2, 1) USC_WF(1, 1)
    USC_WF(4, 3)
#undef USC_WF
    if (S > 1)
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(stream_grid(numel, 256)), dim3(256), 0, st, (const float*)ws, (int)S,
                         numel, (int)accumulate, dW);
    USC_CHECK_LAUNCH("usc_spconv_wgrad");
    return USC_OK;
  }
#endif
  const int NB = pick_nb(cout);
  const bool aligned = (cin % 32 == 0) && (cout % (NB * 32) == 0);
  const int S = wgrad_splits(K, cin, cout, NB, n_rows);
  USC_REQUIRE(ws_bytes >= (int64_t)S * numel * 4, "usc_spconv_wgrad: workspace too small");
  p.S = S;
  dim3 grid((unsigned)(K * S), (unsigned)ceil_div(cin, 32), (unsigned)ceil_div(cout, NB * 32));
  const size_t lds = (size_t)3 * NB * 16 * 64 * sizeof(float);
#define USC_WG(NBv)                                                                          \
  if (aligned) hipLaunchKernelGGL((wgrad_kernel<NBv, true>), grid, dim3(256), lds, st, p);  \
  else hipLaunchKernelGGL((wgrad_kernel<NBv, false>), grid, dim3(256), lds, st, p);
  switch (NB) {
    case 1: USC_WG(1) break;
    case 2: USC_WG(2) break;
    case 3: USC_WG(3) break;
    default: USC_WG(4) break;
  }
#undef USC_WG
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(stream_grid(numel, 256)), dim3(256), 0, st, (const float*)ws, S, numel,
                     (int)accumulate, dW);
  USC_CHECK_LAUNCH("usc_spconv_wgrad");
  return USC_OK;
}

int32_t usc_spconv_wgrad_group_max(void) { return kMaxWgradGroup; }

int usc_spconv_wgrad_group_ok(int32_t R, int32_t cin, int32_t cout, int32_t K, int64_t n_rows) {
  // the grouped form covers what wgrad_full_kernel covers, and only where ONE slice per problem fills the chip
  if (R < 2 || R > kMaxWgradGroup || cin % 32 || cout % 32 || K < 1 || n_rows <= 0) return 0;
  const int ctiles = cin / 32, cb = cout / 32;
  const int NBf = wgrad_full_nb(cb);
  if (NBf == 3 && ctiles % 3 != 0) return 0;
  const int CT = wgrad_full_ct(ctiles, NBf);
  const int64_t blocks = (int64_t)R * K * (ctiles / CT) * (cb / NBf);
  // per workgroup the whole pair list of one offset: bounded so that the longest chain stays short next to the launch
  return blocks >= 256 && n_rows / K <= 2048 ? 1 : 0;
}

int usc_spconv_wgrad_group(int32_t R, const float* const* a, const float* const* b, float* const* dW, int32_t cin,
                           int32_t cout, int32_t K, const int32_t* a_idx, const int32_t* b_idx, const int64_t* koff,
                           int64_t n_rows, int32_t accumulate, usc_stream_t s) {
  USC_REQUIRE(R >= 1 && R <= kMaxWgradGroup, "usc_spconv_wgrad_group: 1 <= R <= %d problems per launch", kMaxWgradGroup);
  USC_REQUIRE(a && b && dW, "usc_spconv_wgrad_group: null pointer table");
  USC_REQUIRE(cin >= 32 && cin % 32 == 0 && cout >= 32 && cout % 32 == 0 && K >= 1 && n_rows >= 0,
              "usc_spconv_wgrad_group: channels must be multiples of 32");
  USC_REQUIRE((a_idx && b_idx && koff) || (!a_idx && K == 1), "usc_spconv_wgrad_group: pair lists required for K>1");
  const int ctiles = cin / 32, cb = cout / 32;
  const int NBf = wgrad_full_nb(cb);
  USC_REQUIRE(!(NBf == 3 && ctiles % 3 != 0), "usc_spconv_wgrad_group: unsupported channel pair %d -> %d", cin, cout);
  if (n_rows == 0) return USC_OK;
  WgradGroup g{};
  g.R = R;
  for (int r = 0; r < R; ++r) {
    USC_REQUIRE(a[r] && b[r] && dW[r], "usc_spconv_wgrad_group: null pointer in problem %d", r);
    g.a[r] = a[r]; g.b[r] = b[r]; g.dW[r] = dW[r];
  }
  WgradParams p{};
  p.a_idx = a_idx; p.b_idx = b_idx; p.koff = koff; p.n_rows = n_rows; p.cin = cin; p.cout = cout; p.K = K;
  p.S = 1; p.accumulate = accumulate; p.direct = g.dW[0]; p.a = g.a[0]; p.b = g.b[0];
  const int CT = wgrad_full_ct(ctiles, NBf);
  hipStream_t st = as_stream(s);
  dim3 grid((unsigned)(R * K), (unsigned)(ctiles / CT), (unsigned)(cb / NBf));
  const size_t lds = (size_t)4 * NBf * 16 * 64 * sizeof(float);
#define USC_WF(C, N) if (CT == C && NBf == N) { \
    static bool attr_set = false; auto kfn = wgrad_full_kernel<C, N>; \
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); attr_set = true; } \
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, st, p, g); }
  USC_WF(3, 3) USC_WF(2, 3) USC_WF(1, 3) USC_WF(2, 4) USC_WF(1, 4) USC_WF(4, 2) USC_WF(3, 2) USC_WF(2, 2) USC_WF(1, 2) USC_WF(4, 1) USC_WF(3, 1) USC_WF(2, 1) USC_WF(1, 1)
#undef USC_WF
  USC_CHECK_LAUNCH("usc_spconv_wgrad_group");
  return USC_OK;
}

}  // extern "C"
