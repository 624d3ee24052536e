// Sparse convolution, table form, large maps: accumulators in REGISTERS over all offsets of a mask-sorted 32-row tile.
//
// The tile-compacted kernel (spconv.hip) packs the real (in, out) pairs of an offset into groups of 32 and pays for it
// with an ordered read-modify-write of every group's 32 x BN results into an LDS tile — 19 % of the launch once the
// operand loads are out of the way (profiles/r04_compact_skeleton_sq.txt) — and with a tile-level tail (four 148-row
// tiles per CU).  Here a wave owns 32 output rows that the mask sort (usc_rowsort_build) has made alike in their
// neighbour masks, walks the offsets the tile has at all, and keeps the 32 x BN results in registers from the first
// offset to the last: no flush, no ticket, no LDS tile, work units of 32 rows.  The price is padding inside the tile
// (0.757 real pairs per issued slot against 0.835).  Operand path = the compacted kernel's: gathered rows three quads
// ahead, packed weight slices (one 16-byte load per lane, quad and accumulator) one quad ahead, rings run across offsets.
// Every output element is reduced by one lane over k ascending, channel ascending: the same bits as the mask-sorted and
// the row-order kernels.
#include "common.h"

namespace usc {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

constexpr int kRegK = 27;
constexpr int kRegZero = 4096;
__device__ float r_zero_row[kRegZero + 8];
constexpr int kRegWaves = 4;

struct RegaccParams {
  const float* in;          // [n_in, cin]
  const float* Wp;          // packed weights (regacc_pack_kernel)
  const int32_t* nbr;       // [K, n_out]
  const int32_t* perm;      // sorted position -> row
  const uint32_t* tmask;    // per 32-row sorted tile: OR of the rows' neighbour masks
  const float* bias;
  float* out;               // [n_out, cout]
  int64_t n_out;
  int cin, cout, K, accumulate;
};

// Wp[cb][k][q][nb][lane = 32h + i][j] = W[k][8q + 4h + j][cb*32*NB + NB*i + nb]   (the compacted kernel's operand order;
// wt = 1: from the forward weights [K, cout, cin], offsets mirrored — the input gradient of a stride-1 convolution)
__global__ void regacc_pack_kernel(const float* __restrict__ W, int K, int cin, int cout, int NB, int wt, float* __restrict__ out) {
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

__device__ inline int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

template <int NB>
__global__ __launch_bounds__(64 * kRegWaves, 4) void gather_gemm_regacc_kernel(RegaccParams p) {
  constexpr int BN = NB * 32;
  __shared__ int32_t inrow[kRegWaves][kRegK][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int cin = p.cin, cout = p.cout, K = p.K;
  const int n0 = blockIdx.y * BN;
  const int64_t ntiles = (p.n_out + 31) >> 5;
  const int64_t tile = (int64_t)blockIdx.x * kRegWaves + wave;
  if (tile >= ntiles) return;              // (no workgroup barrier below: waves are independent)
  const int64_t srow = tile * 32 + i;
  const int my_row = srow < p.n_out ? p.perm[srow] : -1;
  const uint32_t mask = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.tmask[tile]);
  // neighbour rows of the tile for the offsets it has: [k][i] in this wave's LDS slice (all loads in flight together)
  for (int kk = h; kk < K; kk += 2) {
    int v = -1;
    if (((mask >> kk) & 1u) && my_row >= 0) v = p.nbr[(int64_t)kk * p.n_out + my_row];
    inrow[wave][kk][i] = v;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();

  struct Item { const float* arow; const float* wk; int k; };
  const int nq = cin >> 3;
  const float* wp_cb = p.Wp + (int64_t)blockIdx.y * K * cin * BN;
  uint32_t rest = mask;
  auto make_item = [&]() -> Item {      // the next offset of the tile (ascending k), or a dummy past the end
    Item st;
    if (!rest) { st.k = -1; st.arow = r_zero_row + 4 * h; st.wk = wp_cb; return st; }
    const int k = __builtin_ctz(rest);
    rest &= rest - 1u;
    const int v = inrow[wave][k][i];
    st.k = k;
    st.arow = (v >= 0 ? p.in + (int64_t)v * cin : r_zero_row) + 4 * h;
    st.wk = wp_cb + (int64_t)k * cin * BN;
    return st;
  };
  constexpr int kDA = 3, kDB = 1, kRA = 4, kRB = 2;
  float4 ra[kRA];
  float4 rb[kRB][NB];
  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
  auto load_bq = [&](float4 (&dst)[NB], const float* wq) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) dst[nb] = *reinterpret_cast<const float4*>(wq + nb * 256 + lane * 4);
  };

  Item cur = make_item();
  Item nxt = make_item();
  if (cur.k >= 0) {
    const float* pa = cur.arow;
    const float* pb = cur.wk;
    int qa = 0, qb = 0;
    auto advance_a = [&]() { pa += 8; ++qa; if (qa == nq) { pa = nxt.arow; qa = 0; } };
    auto advance_b = [&]() { pb += NB * 256; ++qb; if (qb == nq) { pb = nxt.wk; qb = 0; } };
#pragma unroll
    for (int q = 0; q < kDA; ++q) { ra[q % kRA] = *reinterpret_cast<const float4*>(pa); advance_a(); }
#pragma unroll
    for (int q = 0; q < kDB; ++q) { load_bq(rb[q % kRB], pb); advance_b(); }
    int q0 = 0;
    while (true) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ra[(u + kDA) % kRA] = *reinterpret_cast<const float4*>(pa); advance_a();
        load_bq(rb[(u + kDB) % kRB], pb); advance_b();
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
        if (nxt.k < 0) break;
        cur = nxt;
        nxt = make_item();
        q0 = 0;
      }
    }
  }
  // ---- epilogue: row (reg, h) of the tile is sorted position tile*32 + acc_row(reg, h); a lane holds NB adjacent columns
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int orow = __shfl(my_row, acc_row(r, h), 64);
    if (orow < 0) continue;
    float v[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) v[nb] = acc[nb][r];
    float* dst = p.out + (int64_t)orow * cout + n0 + NB * i;
    if (p.bias) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) v[nb] += p.bias[n0 + NB * i + nb];
    }
    if (p.accumulate) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) v[nb] += dst[nb];
    }
    if (NB == 4) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1 % NB], v[2 % NB], v[3 % NB]);
    else if (NB == 2) *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1 % NB]);
    else {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) dst[nb] = v[nb];
    }
  }
}

int regacc_nb(int cout) {
  const int cb = cout / 32;
  return (cb % 3 == 0) ? 3 : (cb % 2 == 0 ? 2 : 1);
}

}  // namespace
}  // namespace usc

using namespace usc;

extern "C" {

int32_t usc_spconv_regacc_ok(int64_t n_out, int32_t cin, int32_t cout, int32_t K) {
  return (K > 1 && K <= kRegK && n_out >= 1 && cin >= 32 && cin % 32 == 0 && cin <= kRegZero && cout >= 32 && cout % 32 == 0) ? 1 : 0;
}

int64_t usc_spconv_regacc_ws_bytes(int32_t cin, int32_t cout, int32_t K) { return (int64_t)K * cin * cout * 4; }

int usc_spconv_regacc_gemm(const float* in, int64_t n_in, int32_t cin, const float* W, int32_t K, int32_t cout,
                           const int32_t* nbr, const int32_t* perm, const uint32_t* tile_mask, int64_t n_out,
                           const float* bias, float* out, int32_t accumulate, int32_t w_transposed, void* ws,
                           int64_t ws_bytes, usc_stream_t s) {
  (void)n_in;
  USC_REQUIRE(usc_spconv_regacc_ok(n_out, cin, cout, K), "usc_spconv_regacc_gemm: unsupported shape");
  USC_REQUIRE(in && W && nbr && perm && tile_mask && out && ws, "usc_spconv_regacc_gemm: null pointer");
  USC_REQUIRE(ws_bytes >= usc_spconv_regacc_ws_bytes(cin, cout, K), "usc_spconv_regacc_gemm: workspace too small");
  hipStream_t st = as_stream(s);
  const int NB = regacc_nb(cout);
  hipLaunchKernelGGL(regacc_pack_kernel, dim3(stream_grid((int64_t)K * cin * cout, 256)), dim3(256), 0, st, W, (int)K, (int)cin,
                     (int)cout, NB, (int)w_transposed, (float*)ws);
  RegaccParams p{};
  p.in = in; p.Wp = (const float*)ws; p.nbr = nbr; p.perm = perm; p.tmask = tile_mask; p.bias = bias; p.out = out;
  p.n_out = n_out; p.cin = cin; p.cout = cout; p.K = K; p.accumulate = accumulate;
  const int64_t ntiles = (n_out + 31) >> 5;
  dim3 grid((unsigned)ceil_div(ntiles, (int64_t)kRegWaves), (unsigned)(cout / (NB * 32)));
  if (NB == 3) hipLaunchKernelGGL(gather_gemm_regacc_kernel<3>, grid, dim3(64 * kRegWaves), 0, st, p);
  else if (NB == 2) hipLaunchKernelGGL(gather_gemm_regacc_kernel<2>, grid, dim3(64 * kRegWaves), 0, st, p);
  else hipLaunchKernelGGL(gather_gemm_regacc_kernel<1>, grid, dim3(64 * kRegWaves), 0, st, p);
  USC_CHECK_LAUNCH("usc_spconv_regacc_gemm");
  return USC_OK;
}

}  // extern "C"
