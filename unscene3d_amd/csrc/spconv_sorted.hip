// spconv_sorted.hip — sparse convolution over MASK-SORTED output rows (gfx950).
//
// Why: with surface voxels only ~13 of 27 neighbours exist, so a 32-row MFMA tile of consecutive rows is ~50 %
// padding (measured 0.488 useful pair slots on the 150 k-voxel scene).  If the output rows are grouped by their
// neighbour BITMASK, the rows of one tile want (almost) the same offsets and a tile simply skips the offsets none
// of its rows has: 0.757 useful slots with a one-pass bucket sort on the 12 edge-offset bits (0.785 for a full
// sort), at ANY map size — the tile-compacted kernel of spconv.hip needs >= 24 k rows to fill the chip.
// Accumulators stay in registers for the whole reduction (no LDS flush, no ordering ticket), which frees the LDS
// for the weights: a workgroup walks the offsets of its rows together and stages each 32-channel weight chunk
// ONCE for its 4-8 tiles; B operands are then LDS reads, not texture-path loads.
//
// The result does not depend on the row permutation: every output element is reduced over (k ascending, channel
// ascending) by one lane, whatever tile its row landed in.  The bucket sort may therefore use atomics.
//
// Reference semantics: MinkowskiConvolution forward / input-gradient on a stride-1 (or k2/s2 child-table) kernel
// map, models/res16unet.py:224-297 via MinkowskiEngine 0.5.4 (un-vendored).
#include "common.h"

namespace usc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

void launch_group_reduce(const float* partial, int G, int64_t numel4, int cout, const float* bias, int accumulate,
                         float* out, hipStream_t st);   // spconv.hip

namespace {

constexpr int kKeyBits = 12;
#ifndef USC_SORT_ROWS
#define USC_SORT_ROWS 2   /* rows per thread in the bucket-sort kernels (workgroup = 256 * USC_SORT_ROWS rows) */
#endif
constexpr int kSortRows = USC_SORT_ROWS;
constexpr int kBins = 1 << kKeyBits;
constexpr int kZeroFloats = 4096;
__device__ float s_zero_row[kZeroFloats + 8];

__device__ inline int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// ---------------------------------------------------------------------------------------------------
// row masks, bucket sort, tile masks
struct KeySel { int bit[kKeyBits]; int n; };

// Row masks + bucket histogram.  Surface voxels put most rows into a few dozen of the 4096 buckets, so global atomics
// per row serialise (45 us for 148 k rows); each workgroup (1024 rows) first counts in an LDS histogram and then
// adds only its non-empty buckets to the global one.
__global__ __launch_bounds__(256) void rowmask_hist_kernel(const int32_t* __restrict__ nbr, int K, int64_t n,
                                                          KeySel sel, uint32_t* __restrict__ mask,
                                                          uint32_t* __restrict__ bins) {
  __shared__ uint32_t lh[kBins];
  for (int e = threadIdx.x; e < kBins; e += 256) lh[e] = 0u;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * (256 * kSortRows);
#pragma unroll
  for (int u = 0; u < kSortRows; ++u) {
    const int64_t r = base + u * 256 + threadIdx.x;
    if (r < n) {
      uint32_t m = 0;
      for (int k = 0; k < K; ++k) m |= (nbr[(int64_t)k * n + r] >= 0 ? 1u : 0u) << k;
      mask[r] = m;
      uint32_t key = 0;
      for (int b = 0; b < sel.n; ++b) key |= ((m >> sel.bit[b]) & 1u) << b;
      atomicAdd(&lh[key], 1u);
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < kBins; e += 256) {
    const uint32_t c = lh[e];
    if (c) atomicAdd(&bins[e], c);
  }
}

// exclusive scan of the kBins bucket counts -> cursors (one workgroup)
__global__ __launch_bounds__(1024) void bins_scan_kernel(const uint32_t* __restrict__ bins, uint32_t* __restrict__ cursor) {
  __shared__ uint32_t part[1024];
  constexpr int per = kBins / 1024;
  uint32_t v[per], s = 0;
#pragma unroll
  for (int j = 0; j < per; ++j) { v[j] = bins[threadIdx.x * per + j]; s += v[j]; }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const uint32_t t = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t base = part[threadIdx.x] - s;
#pragma unroll
  for (int j = 0; j < per; ++j) { cursor[threadIdx.x * per + j] = base; base += v[j]; }
}

// Placement: per workgroup (1024 rows) an LDS count per bucket gives every row its rank inside the workgroup's
// share of the bucket; one global atomic per (workgroup, non-empty bucket) reserves the share.
__global__ __launch_bounds__(256) void bucket_place_kernel(const uint32_t* __restrict__ mask, int64_t n, KeySel sel,
                                                          uint32_t* __restrict__ cursor, int32_t* __restrict__ perm) {
  __shared__ uint32_t lh[kBins];
  for (int e = threadIdx.x; e < kBins; e += 256) lh[e] = 0u;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * (256 * kSortRows);
  uint32_t key[kSortRows], rank[kSortRows];
#pragma unroll
  for (int u = 0; u < kSortRows; ++u) {
    const int64_t r = base + u * 256 + threadIdx.x;
    key[u] = 0; rank[u] = 0;
    if (r < n) {
      const uint32_t m = mask[r];
      for (int b = 0; b < sel.n; ++b) key[u] |= ((m >> sel.bit[b]) & 1u) << b;
      rank[u] = atomicAdd(&lh[key[u]], 1u);
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < kBins; e += 256) {
    const uint32_t c = lh[e];
    if (c) lh[e] = atomicAdd(&cursor[e], c);     // count -> start of this workgroup's share
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kSortRows; ++u) {
    const int64_t r = base + u * 256 + threadIdx.x;
    if (r < n) perm[lh[key[u]] + rank[u]] = (int32_t)r;
  }
}

__global__ __launch_bounds__(256) void tile_mask_kernel(const uint32_t* __restrict__ mask, const int32_t* __restrict__ perm,
                                                       int64_t n, uint32_t* __restrict__ tmask) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t m = j < n ? mask[perm[j]] : 0u;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) m |= __shfl_xor(m, o, 64);   // OR over each aligned group of 32 lanes
  if ((threadIdx.x & 31) == 0 && j < n) tmask[j >> 5] = m;
}

// ---------------------------------------------------------------------------------------------------
struct SortedParams {
  const float* in;        // [n_in, cin]
  const float* W;         // [K, cin, cout]  natural layout
  const int32_t* nbr;     // [K, n_out]
  const int32_t* perm;    // [n_out] rows grouped by neighbour mask
  const uint32_t* tmask;  // [ceil(n_out/32)] OR of the row masks of each sorted tile
  const float* bias;
  float* out;             // [n_out, cout]  or partial [G, n_out, cout]
  int64_t n_out;
  int cin, cout, K, G, accumulate;
  int wt;                 // 1: W is the FORWARD conv's [K, cout, cin]; use W'[k][c][n] = W[K-1-k][n][c] (input gradient)
};

// One wave = one sorted tile of 32 rows x (32*NB) columns, accumulators in registers over the whole reduction.
// The WAVES waves of a workgroup walk the union of their offsets together; per (offset, 32-channel chunk) step
// the chunk W[k][c0:c0+32][n0:n0+BN] is staged once into LDS (double-buffered: the global loads of step s+1 are
// issued before the matrix-core work of step s and written to LDS after it; one barrier per step).  A wave whose
// tile has no row with offset k skips the step's compute (it still helps staging).  Gathered rows run through
// a register ring three quads ahead, across step and offset boundaries.
template <int NB, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES == 8 ? 4 : 3) void gather_gemm_sorted_kernel(SortedParams p) {
  constexpr int BN = NB * 32;
  constexpr int NT = 64 * WAVES;
  constexpr int kF4 = 8 * BN;                        // float4s per staged chunk (32 rows x BN floats)
  constexpr int kLd = (kF4 + NT - 1) / NT;           // staging float4s per thread
  __shared__ __attribute__((aligned(16))) float wbuf[2][32 * BN];
  __shared__ int32_t inrow[WAVES][32][33];           // [wave][k][pair i] (+1 pad)
  __shared__ uint32_t wmask[WAVES];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.y * BN;
  const int cin = p.cin, cout = p.cout, K = p.K;
  const int64_t ntiles = (p.n_out + 31) >> 5;
  const int64_t tile = (int64_t)blockIdx.x * WAVES + wave;
  const int Kg = (K + p.G - 1) / p.G;
  const int k_begin = blockIdx.z * Kg;
  const int k_end = k_begin + Kg < K ? k_begin + Kg : K;
  // (a slice past the last offset has an empty range; the shifts are only taken below 32)
  const uint32_t lo = k_begin >= 32 ? 0xffffffffu : ((1u << k_begin) - 1u);
  const uint32_t hi = k_end >= 32 ? 0xffffffffu : ((1u << k_end) - 1u);
  const uint32_t range = k_begin < k_end ? (hi & ~lo) : 0u;

  const int64_t srow = tile * 32 + i;
  const int my_row = (tile < ntiles && srow < p.n_out) ? p.perm[srow] : -1;
  uint32_t my_mask = (tile < ntiles) ? (p.tmask[tile] & range) : 0u;
  my_mask = __builtin_amdgcn_readfirstlane(my_mask);
  if (lane == 0) wmask[wave] = my_mask;
  // neighbour rows of this tile for every offset it needs: [k][i] in LDS (one gather pass, all loads in flight)
  for (int kk = h; kk < K; kk += 2) {
    int v = -1;
    if (((my_mask >> kk) & 1u) && my_row >= 0) v = p.nbr[(int64_t)kk * p.n_out + my_row];
    inrow[wave][kk][i] = v;
  }
  __syncthreads();
  uint32_t block_mask = 0;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) block_mask |= wmask[w];
  block_mask = __builtin_amdgcn_readfirstlane(block_mask);

  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  const int nch = cin >> 5;
  // ---- gathered-row stream of this wave: one chunk of 32 channels (4 quads of 8) per compute step, fetched one
  // whole step ahead: quad u of the NEXT compute step is loaded into ring slot u right after quad u of the
  // current step has been issued to the matrix cores.
  auto row_ptr = [&](int k) __attribute__((always_inline)) -> const float* {
    const int v = inrow[wave][k][i];
    return (v >= 0 ? p.in + (int64_t)v * cin : s_zero_row) + 4 * h;
  };
  float4 ra[4], ran[4];   // current compute step's gathered quads / the next compute step's (in flight)
  int ka = my_mask ? __builtin_ctz(my_mask) : 32;   // (ka, cha) = the compute step held in `ran`
  int cha = 0;
  const float* pa = ka < 32 ? row_ptr(ka) : s_zero_row + 4 * h;
#pragma unroll
  for (int u = 0; u < 4; ++u) ran[u] = *reinterpret_cast<const float4*>(pa + 8 * u);

  // ---- weight staging: chunk W[k][c0:c0+32][n0:n0+BN] copied row-major into LDS ([32][BN], one b128 per
  // thread and float4).  Accumulator nb of lane (i,h) owns output column n0 + 32*nb + i, so the B operand of a
  // k-step is wbuf[row][32*nb + i]: a conflict-free ds_read_b32 whose address is ONE lane register plus an
  // immediate — no per-thread address tables (an operand-order LDS layout needed 16 live address registers
  // for its scattered staging writes and pushed the kernel into scratch).
  float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0, w2 = w0, w3 = w0;
  // wt = 1 (input gradient of a stride-1 conv): the chunk is read from the forward weights, transposed and with the
  // offsets mirrored — element (r, n) = W[K-1-k][n0+n][c0+r]; a thread's float4 then runs along r (contiguous in
  // memory) and lands in four LDS rows.  No separate weight-transpose launch per conv.
#define USC_STAGE_LD(U, REG)                                                                   \
  if (U < kLd) {   /* unconditional (index wrapped): a predicated load would force vmcnt(0) at the store */ \
    const int e0 = threadIdx.x + U * NT;                                                       \
    const int e = (kF4 % NT == 0 || e0 < kF4) ? e0 : e0 - kF4;                                 \
    if (wt) {                                                                                  \
      const int nn = e >> 3, r4 = e & 7;                                                       \
      REG = *reinterpret_cast<const float4*>(src + (int64_t)nn * cin + r4 * 4);                \
    } else {                                                                                   \
      const int r = e / (BN / 4), c4 = e - r * (BN / 4);                                       \
      REG = *reinterpret_cast<const float4*>(src + (int64_t)r * cout + c4 * 4);                \
    }                                                                                          \
  }
#define USC_STAGE_ST(U, REG)                                                                   \
  if (U < kLd) {                                                                               \
    const int e = threadIdx.x + U * NT;                                                        \
    if (kF4 % NT == 0 || e < kF4) {                                                            \
      if (wt) {                                                                                \
        const int nn = e >> 3, r4 = e & 7;                                                     \
        float* dstw = wbuf[buf] + (4 * r4) * BN + nn;                                          \
        dstw[0] = REG.x; dstw[BN] = REG.y; dstw[2 * BN] = REG.z; dstw[3 * BN] = REG.w;         \
      } else {                                                                                 \
        reinterpret_cast<float4*>(wbuf[buf])[e] = REG;                                         \
      }                                                                                        \
    }                                                                                          \
  }
  const bool wt = p.wt != 0;
  auto stage_load = [&](int k, int ch) __attribute__((always_inline)) {
    const float* src = wt ? p.W + ((int64_t)(K - 1 - k) * cout + n0) * cin + ch * 32
                          : p.W + ((int64_t)k * cin + ch * 32) * cout + n0;
    USC_STAGE_LD(0, w0) USC_STAGE_LD(1, w1) USC_STAGE_LD(2, w2) USC_STAGE_LD(3, w3)
  };
  auto stage_store = [&](int buf) __attribute__((always_inline)) {
    USC_STAGE_ST(0, w0) USC_STAGE_ST(1, w1) USC_STAGE_ST(2, w2) USC_STAGE_ST(3, w3)
  };
#undef USC_STAGE_LD
#undef USC_STAGE_ST
  static_assert(kLd <= 4, "staging registers");

  // ---- walk the steps (k in block_mask ascending) x (chunks)
  uint32_t rest = block_mask;
  int k = rest ? __builtin_ctz(rest) : 32;
  int ch = 0, step = 0;
  if (k < 32) {
    stage_load(k, 0);
    stage_store(0);
  }
  __syncthreads();
#ifdef USC_ABLATE_SORTED_LOOP   /* developer switch (tools/build_ablate.sh): prologue + epilogue only */
  if (p.cin < 0)
#endif
  while (k < 32) {
    // next step's coordinates
    int k2 = k, ch2 = ch + 1;
    if (ch2 == nch) {
      ch2 = 0;
      const uint32_t r2 = rest & (rest - 1u);
      k2 = r2 ? __builtin_ctz(r2) : 32;
    }
    const bool mine = (my_mask >> k) & 1u;
    if (mine) {
      // Gathers first, weight staging loads second: vector loads complete in order, so the wait for the staged
      // weights at the end of this step then never waits for YOUNGER gathers (the reverse order made the
      // compiler drain the gathers with vmcnt(0) before every barrier).  `ran` was filled one compute step ago.
#pragma unroll
      for (int u = 0; u < 4; ++u) ra[u] = ran[u];
      ++cha;
      if (cha == nch) {
        cha = 0;
        const uint32_t mrest = (ka >= 31) ? 0u : (my_mask & ~((2u << ka) - 1u));
        ka = mrest ? __builtin_ctz(mrest) : 32;
        pa = ka < 32 ? row_ptr(ka) : s_zero_row + 4 * h;
      } else {
        pa += 32;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) ran[u] = *reinterpret_cast<const float4*>(pa + 8 * u);
    }
    if (k2 < 32) stage_load(k2, ch2);
    if (mine) {
      const float* wl = wbuf[step & 1] + (4 * h) * BN + i;   // + (8u + j) * BN + 32 * nb
      // rolling single buffer for the B operands: the slot of k-step j is refilled for the next quad right after
      // its MFMAs have been issued, so the LDS latency hides behind the other nine MFMAs of the quad
      float bq[4][NB];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bq[j][nb] = wl[j * BN + 32 * nb];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 a4 = ra[u];
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = MFMA32(av[j], bq[j][nb], acc[nb]);
          if (u < 3) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bq[j][nb] = wl[(8 * (u + 1) + j) * BN + 32 * nb];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (k2 < 32) stage_store((step + 1) & 1);
    __syncthreads();
    if (ch2 == 0) rest &= rest - 1u;
    k = k2; ch = ch2; ++step;
  }

  // ---- write the tile (rows scattered back through perm)
  if (tile >= ntiles) return;
#ifdef USC_ABLATE_SORTED_STORE  /* developer switch: no tile write-back (one word, so that the accumulators stay live) */
  if (p.cin > 0) { if (lane == 0 && acc[0][0] == 123.456f) p.out[0] = acc[NB - 1][15]; return; }
#endif
  float* outp = p.out;
  if (p.G > 1) outp += (int64_t)blockIdx.z * p.n_out * cout;
  const bool direct = p.G == 1;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int orow = __shfl(my_row, acc_row(r, h), 64);
    if (orow < 0) continue;
    float v[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) v[nb] = acc[nb][r];
    float* dst = outp + (int64_t)orow * cout + n0 + i;   // accumulator nb -> column n0 + 32*nb + i
    if (direct) {
      if (p.bias) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) v[nb] += p.bias[n0 + 32 * nb + i];
      }
      if (p.accumulate) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) v[nb] += dst[32 * nb];
      }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) dst[32 * nb] = v[nb];
  }
}

struct SortedPlan { int NB, WAVES, G; };

// Tile width (NB blocks of 32 output columns per wave) and offset split (G partial-sum slices) of one launch.
// Measured on the bench scene's 507 ... 40 421-row levels (tools/sorted_plan_sweep.py -> profiles/r02_sorted_plan.txt):
// the tile width hardly matters at a given G; what matters is (a) enough waves that every SIMD holds two or three
// (a lone wave runs its chain of matrix-core steps at ~60 %: nothing overlaps its barriers and LDS reads) against
// (b) one more [n_out, cout] slice written and read back per extra G.  (a) wins up to ~2 waves per SIMD on the
// <= 2 222-row levels and ~3 on the larger ones, where a slice is cheap relative to the chain.
SortedPlan plan_sorted(int64_t n_out, int cin, int cout, int K) {
  const int cb = cout / 32;
  int nb = 1;
  if (cb % 4 == 0) nb = 4;
  else if (cb % 3 == 0) nb = 3;
  else if (cb % 2 == 0) nb = 2;
  // USC3D_SORTED_TUNE=1: tools/sorted_plan_sweep.py forces (NB, G) per launch through the environment
  static const bool tune = getenv("USC3D_SORTED_TUNE") != nullptr;
  if (tune) {
    const char* e_nb = getenv("USC3D_SORTED_NB");
    const char* e_g = getenv("USC3D_SORTED_G");
    const int f_nb = e_nb ? atoi(e_nb) : 0, f_g = e_g ? atoi(e_g) : 0;
    if (f_nb >= 1 && f_nb <= 4 && cb % f_nb == 0 && f_g >= 1) {
      const int64_t Kg = ceil_div((int64_t)K, (int64_t)(f_g < K ? f_g : K));
      return SortedPlan{f_nb, f_nb == 4 ? 4 : 8, (int)ceil_div((int64_t)K, Kg)};
    }
  }
  SortedPlan pl{nb, nb == 4 ? 4 : 8, 1};
  const int64_t ntiles = ceil_div(n_out, 32);
  const int64_t waves = ntiles * (cb / nb);
  const int64_t kTarget = n_out >= 4096 ? 3072 : 2048;
  if (waves > 0 && waves < kTarget) {   // (an empty map plans to nothing)
    int64_t G = ceil_div(kTarget, waves);
    if (G > K) G = K;
    // no empty slices: G = 20 on 27 offsets means 2 per slice, which 14 slices cover
    const int64_t Kg = ceil_div((int64_t)K, G);
    pl.G = (int)ceil_div((int64_t)K, Kg);
  }
  return pl;
}

KeySel key_selection(int K) {
  KeySel s{};
  if (K == 27) {
    // the 12 edge offsets of the 3x3x3 stencil (exactly one zero coordinate; occupancy 0.42-0.54 on surface
    // voxels): the most informative 12 of the 27 bits, 0.757 tile efficiency vs 0.785 for a full 27-bit sort
    const int e[12] = {1, 3, 5, 7, 9, 11, 15, 17, 19, 21, 23, 25};
    for (int b = 0; b < 12; ++b) s.bit[b] = e[b];
    s.n = 12;
  } else {
    s.n = K < kKeyBits ? K : kKeyBits;
    for (int b = 0; b < s.n; ++b) s.bit[b] = b;
  }
  return s;
}

}  // namespace
}  // namespace usc

using namespace usc;

extern "C" {

int64_t usc_rowsort_ws_bytes(int32_t K, int64_t n_out) {
  (void)K;
  return align_up(n_out * 4, 256) + 2 * kBins * 4;
}

int usc_rowsort_build(const int32_t* nbr, int32_t K, int64_t n_out, int32_t* perm, uint32_t* tile_mask, void* ws,
                      int64_t ws_bytes, usc_stream_t s) {
  USC_REQUIRE(K >= 1 && K <= 32 && n_out >= 0, "usc_rowsort_build: K must be in 1..32");
  if (n_out == 0) return USC_OK;
  USC_REQUIRE(nbr && perm && tile_mask && ws, "usc_rowsort_build: null pointer");
  USC_REQUIRE(ws_bytes >= usc_rowsort_ws_bytes(K, n_out), "usc_rowsort_build: workspace too small");
  hipStream_t st = as_stream(s);
  uint32_t* mask = (uint32_t*)ws;
  uint32_t* bins = (uint32_t*)((char*)ws + align_up(n_out * 4, 256));
  uint32_t* cursor = bins + kBins;
  const KeySel sel = key_selection(K);
  (void)hipMemsetAsync(bins, 0, kBins * 4, st);
  const unsigned nb = (unsigned)ceil_div(n_out, 256);
  const unsigned nb4 = (unsigned)ceil_div(n_out, 256 * kSortRows);
  hipLaunchKernelGGL(rowmask_hist_kernel, dim3(nb4), dim3(256), 0, st, nbr, (int)K, n_out, sel, mask, bins);
  hipLaunchKernelGGL(bins_scan_kernel, dim3(1), dim3(1024), 0, st, (const uint32_t*)bins, cursor);
  hipLaunchKernelGGL(bucket_place_kernel, dim3(nb4), dim3(256), 0, st, (const uint32_t*)mask, n_out, sel, cursor, perm);
  hipLaunchKernelGGL(tile_mask_kernel, dim3(nb), dim3(256), 0, st, (const uint32_t*)mask, (const int32_t*)perm, n_out,
                     tile_mask);
  USC_CHECK_LAUNCH("usc_rowsort_build");
  return USC_OK;
}

int64_t usc_spconv_sorted_ws_bytes(int64_t n_out, int32_t cin, int32_t cout, int32_t K) {
  if (cin % 32 || cout % 32 || K < 1) return 0;
  const SortedPlan pl = plan_sorted(n_out, cin, cout, K);
  return pl.G > 1 ? (int64_t)pl.G * n_out * cout * 4 : 0;
}

int usc_spconv_sorted_gemm(const float* in, int64_t n_in, int32_t cin, const float* W, int32_t K, int32_t cout,
                           const int32_t* nbr, const int32_t* perm, const uint32_t* tile_mask, int64_t n_out,
                           const float* bias, float* out, int32_t accumulate, int32_t w_transposed, void* ws,
                           int64_t ws_bytes, usc_stream_t s) {
  return usc_spconv_sorted_gemm_ex(in, n_in, cin, W, K, cout, nbr, perm, tile_mask, n_out, bias, out, accumulate,
                                   w_transposed, ws, ws_bytes, nullptr, s);
}

int usc_spconv_sorted_gemm_ex(const float* in, int64_t n_in, int32_t cin, const float* W, int32_t K, int32_t cout,
                              const int32_t* nbr, const int32_t* perm, const uint32_t* tile_mask, int64_t n_out,
                              const float* bias, float* out, int32_t accumulate, int32_t w_transposed, void* ws,
                              int64_t ws_bytes, int32_t* slices_left, usc_stream_t s) {
  if (slices_left) *slices_left = 0;
  USC_REQUIRE(n_in >= 0 && n_out >= 0 && K >= 1 && K <= 32, "usc_spconv_sorted_gemm: bad sizes");
  USC_REQUIRE(cin >= 32 && cin % 32 == 0 && cout >= 32 && cout % 32 == 0 && cin <= kZeroFloats,
              "usc_spconv_sorted_gemm: channels must be multiples of 32 (cin <= 4096)");
  if (n_out == 0) return USC_OK;
  USC_REQUIRE(in && W && nbr && perm && tile_mask && out, "usc_spconv_sorted_gemm: null pointer");
  const SortedPlan pl = plan_sorted(n_out, cin, cout, K);
  SortedParams p{};
  p.in = in; p.W = W; p.nbr = nbr; p.perm = perm; p.tmask = tile_mask; p.bias = bias; p.out = out;
  p.n_out = n_out; p.cin = cin; p.cout = cout; p.K = K; p.G = pl.G; p.accumulate = accumulate; p.wt = w_transposed;
  if (pl.G > 1) {
    USC_REQUIRE(ws && ws_bytes >= (int64_t)pl.G * n_out * cout * 4, "usc_spconv_sorted_gemm: workspace too small");
    p.out = (float*)ws;
  }
  hipStream_t st = as_stream(s);
  const int64_t ntiles = ceil_div(n_out, 32);
  dim3 grid((unsigned)ceil_div(ntiles, pl.WAVES), (unsigned)(cout / (pl.NB * 32)), (unsigned)pl.G);
  if (pl.NB == 4) hipLaunchKernelGGL((gather_gemm_sorted_kernel<4, 4>), grid, dim3(256), 0, st, p);
  else if (pl.NB == 3) hipLaunchKernelGGL((gather_gemm_sorted_kernel<3, 8>), grid, dim3(512), 0, st, p);
  else if (pl.NB == 2) hipLaunchKernelGGL((gather_gemm_sorted_kernel<2, 8>), grid, dim3(512), 0, st, p);
  else hipLaunchKernelGGL((gather_gemm_sorted_kernel<1, 8>), grid, dim3(512), 0, st, p);
  if (pl.G > 1) {
    // slices_left: the caller sums the G slices [G][n_out][cout] at the start of `ws` itself (fused with what it does
    // next: usc_bn_tile_forward / _backward) — `out`, `bias` and `accumulate` are then the caller's business too
    if (slices_left && !bias) *slices_left = pl.G;
    else launch_group_reduce((const float*)ws, pl.G, n_out * cout / 4, (int)cout, bias, (int)accumulate, out, st);
  }
  USC_CHECK_LAUNCH("usc_spconv_sorted_gemm");
  return USC_OK;
}

}  // extern "C"
