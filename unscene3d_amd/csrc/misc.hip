// misc.hip — remaining operators of the path:
//   exact 1-nearest-neighbour (scipy.spatial.KDTree.query(k=1)),
//   eps-ball connected components (sklearn DBSCAN(eps, min_samples=1)),
//   tri-plane projection of sparse voxels (custom_cuda_utils.project_sparse_voxels_to_planes{,_backward}).
#include "common.h"

namespace usc {

// ---------------------------------------------------------------------------
// 1-NN: every thread owns one query, reference points stream through LDS in tiles.  Distances in
// f64 (inputs are f32, so differences and squares are exact up to the final sums — the same
// ordering scipy's f64 KD-tree sees); ties resolve to the lowest reference index.
constexpr int kKnnTile = 1024;

__global__ __launch_bounds__(256) void knn1_kernel(const float* __restrict__ q, int64_t nq, const float* __restrict__ r,
                                                  int64_t nr, int64_t* __restrict__ idx, float* __restrict__ dist2) {
  __shared__ float tx[kKnnTile], ty[kKnnTile], tz[kKnnTile];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool act = i < nq;
  const double qx = act ? (double)q[i * 3 + 0] : 0.0, qy = act ? (double)q[i * 3 + 1] : 0.0,
               qz = act ? (double)q[i * 3 + 2] : 0.0;
  double best = 1e300;
  int64_t bi = -1;
  for (int64_t t0 = 0; t0 < nr; t0 += kKnnTile) {
    const int cnt = (int)((nr - t0) < kKnnTile ? (nr - t0) : kKnnTile);
    for (int j = threadIdx.x; j < cnt; j += 256) {
      tx[j] = r[(t0 + j) * 3 + 0];
      ty[j] = r[(t0 + j) * 3 + 1];
      tz[j] = r[(t0 + j) * 3 + 2];
    }
    __syncthreads();
    if (act) {
      for (int j = 0; j < cnt; ++j) {
        const double dx = qx - (double)tx[j], dy = qy - (double)ty[j], dz = qz - (double)tz[j];
        const double d = dx * dx + dy * dy + dz * dz;
        if (d < best) { best = d; bi = t0 + j; }   // strict: the lowest index wins a tie
      }
    }
    __syncthreads();
  }
  if (act) {
    idx[i] = bi;
    if (dist2) dist2[i] = (float)best;
  }
}

// ---------------------------------------------------------------------------
// eps-ball connected components by min-label propagation with pointer jumping.
//   label[i] <- min(label[j] : |x_i - x_j| <= eps), then label[i] <- label[label[i]] ... until stable.
// The root of a component is its smallest point index, so ranking the roots in ascending order gives
// sklearn's cluster numbering (clusters are numbered in order of their first point).
__global__ __launch_bounds__(256) void cc_init_kernel(int32_t* __restrict__ label, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) label[i] = (int32_t)i;
}
__global__ __launch_bounds__(256) void cc_step_kernel(const float* __restrict__ xyz, int64_t n, double eps2,
                                                     const int32_t* __restrict__ lin, int32_t* __restrict__ lout,
                                                     int32_t* __restrict__ changed) {
  __shared__ float tx[kKnnTile], ty[kKnnTile], tz[kKnnTile];
  __shared__ int32_t tl[kKnnTile];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool act = i < n;
  const double qx = act ? (double)xyz[i * 3 + 0] : 0.0, qy = act ? (double)xyz[i * 3 + 1] : 0.0,
               qz = act ? (double)xyz[i * 3 + 2] : 0.0;
  int32_t best = act ? lin[i] : 0x7fffffff;
  for (int64_t t0 = 0; t0 < n; t0 += kKnnTile) {
    const int cnt = (int)((n - t0) < kKnnTile ? (n - t0) : kKnnTile);
    for (int j = threadIdx.x; j < cnt; j += 256) {
      tx[j] = xyz[(t0 + j) * 3 + 0];
      ty[j] = xyz[(t0 + j) * 3 + 1];
      tz[j] = xyz[(t0 + j) * 3 + 2];
      tl[j] = lin[t0 + j];
    }
    __syncthreads();
    if (act) {
      for (int j = 0; j < cnt; ++j) {
        const double dx = qx - (double)tx[j], dy = qy - (double)ty[j], dz = qz - (double)tz[j];
        if (dx * dx + dy * dy + dz * dz <= eps2 && tl[j] < best) best = tl[j];
      }
    }
    __syncthreads();
  }
  if (act) {
    lout[i] = best;
    if (best != lin[i]) atomicOr(changed, 1);
  }
}
__global__ __launch_bounds__(256) void cc_jump_kernel(int32_t* __restrict__ label, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int32_t l = label[i];
  for (int k = 0; k < 32; ++k) {   // follow the chain to its root (roots satisfy label[r] == r)
    const int32_t p = label[l];
    if (p == l) break;
    l = p;
  }
  label[i] = l;
}
// rank roots: isroot[i] = (label[i] == i); prefix over isroot gives the cluster id of each root
__global__ __launch_bounds__(256) void cc_rootflag_kernel(const int32_t* __restrict__ label, int64_t n,
                                                         int32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) flag[i] = label[i] == (int32_t)i ? 1 : 0;
}
__global__ __launch_bounds__(1024) void cc_scan_kernel(int32_t* __restrict__ flag, int64_t n) {
  // single-block inclusive->exclusive scan (n is at most a few hundred thousand mask points)
  __shared__ int32_t wsum[16];
  __shared__ int32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t c0 = 0; c0 < n; c0 += 1024) {
    const int64_t i = c0 + threadIdx.x;
    const int v = i < n ? flag[i] : 0;
    const int inc = wave_inclusive_scan(v);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    int off = 0, tot = 0;
    for (int k = 0; k < 16; ++k) { if (k < (int)(threadIdx.x >> 6)) off += wsum[k]; tot += wsum[k]; }
    const int carry = carry_s;
    if (i < n) flag[i] = carry + off + inc - v;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void cc_relabel_kernel(const int32_t* __restrict__ label, const int32_t* __restrict__ rank,
                                                        int64_t n, int64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = rank[label[i]];
}

// ---------------------------------------------------------------------------
// tri-plane projection (reference cuda_utils_kernel.cu:371-433 and :496-556)
__global__ __launch_bounds__(256) void project_fwd_kernel(const int32_t* __restrict__ coords, const float* __restrict__ pred,
                                                         const float* __restrict__ tgt, int64_t V, int inst, int xd, int yd,
                                                         int zd, float* pxy, float* pxz, float* pyz, float* txy, float* txz,
                                                         float* tyz, int32_t* nxy, int32_t* nxz, int32_t* nyz) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= V) return;
  const int x = coords[v * 4 + 1], y = coords[v * 4 + 2], z = coords[v * 4 + 3];
  if (x >= xd || y >= yd || z >= zd || x < 0 || y < 0 || z < 0) return;   // dropped, like the reference
  const int64_t sxy = (int64_t)x * yd + y, sxz = (int64_t)x * zd + z, syz = (int64_t)y * zd + z;
  atomicAdd(&nxy[sxy], 1);
  atomicAdd(&nxz[sxz], 1);
  atomicAdd(&nyz[syz], 1);
  for (int p = 0; p < inst; ++p) {
    const float pv = pred[v * inst + p], tv = tgt[v * inst + p];
    atomicAdd(&pxy[sxy * inst + p], pv);
    atomicAdd(&pxz[sxz * inst + p], pv);
    atomicAdd(&pyz[syz * inst + p], pv);
    atomicAdd(&txy[sxy * inst + p], tv);
    atomicAdd(&txz[sxz * inst + p], tv);
    atomicAdd(&tyz[syz * inst + p], tv);
  }
}
__global__ __launch_bounds__(256) void project_bwd_kernel(const int32_t* __restrict__ coords, int64_t V, int inst, int xd,
                                                         int yd, int zd, const float* __restrict__ gxy,
                                                         const float* __restrict__ gxz, const float* __restrict__ gyz,
                                                         float* __restrict__ grad) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= V) return;
  const int x = coords[v * 4 + 1], y = coords[v * 4 + 2], z = coords[v * 4 + 3];
  if (x >= xd || y >= yd || z >= zd || x < 0 || y < 0 || z < 0) return;
  for (int p = 0; p < inst; ++p) {
    const float a = gxy[((int64_t)x * yd + y) * inst + p], b = gxz[((int64_t)x * zd + z) * inst + p],
                c = gyz[((int64_t)y * zd + z) * inst + p];
    const int n = (a != 0.f) + (b != 0.f) + (c != 0.f);
    grad[v * inst + p] = n > 0 ? (a + b + c) / (float)n : 0.f;
  }
}

// ---------------------------------------------------------------------------
// Elastic distortion of a point cloud: xyz += magnitude * trilinear(noise grid)(xyz), in f64 like scipy's
// RegularGridInterpolator (datasets/semseg.py:651-688).  The grid axes come in as the caller's f64 arrays (they are
// np.linspace values); cell index = searchsorted(axis, x) - 1 clipped to [0, dim-2], fraction = (x - a[i]) / (a[i+1] - a[i]).
template <class T>
__global__ __launch_bounds__(256) void elastic_displace_kernel(const T* __restrict__ xyz_in, int64_t n, int32_t stride,
                                                              const float* __restrict__ noise, int dx, int dy, int dz,
                                                              const double* __restrict__ ax, const double* __restrict__ ay,
                                                              const double* __restrict__ az, double magnitude,
                                                              T* __restrict__ xyz_out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double p[3] = {(double)xyz_in[i * stride + 0], (double)xyz_in[i * stride + 1], (double)xyz_in[i * stride + 2]};
  const double* axes[3] = {ax, ay, az};
  const int dims[3] = {dx, dy, dz};
  int c0[3];
  double f[3];
  bool inside = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double* g = axes[a];
    const int d = dims[a];
    inside = inside && p[a] >= g[0] && p[a] <= g[d - 1];
    int k = (int)floor((p[a] - g[0]) / (g[1] - g[0]));     // uniform axis: the guess is off by at most one
    k = k < 0 ? 0 : (k > d - 2 ? d - 2 : k);
    while (k > 0 && p[a] <= g[k]) --k;                      // searchsorted(side='left') - 1
    while (k < d - 2 && p[a] > g[k + 1]) ++k;
    c0[a] = k;
    f[a] = (p[a] - g[k]) / (g[k + 1] - g[k]);
  }
  double disp[3] = {0.0, 0.0, 0.0};
  if (inside) {                                             // bounds_error=0, fill_value=0 outside the grid
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
      const int ox = corner >> 2, oy = (corner >> 1) & 1, oz = corner & 1;
      const double w = (ox ? f[0] : 1.0 - f[0]) * (oy ? f[1] : 1.0 - f[1]) * (oz ? f[2] : 1.0 - f[2]);
      const float* v = noise + ((((int64_t)(c0[0] + ox) * dy + (c0[1] + oy)) * dz) + (c0[2] + oz)) * 3;
      // products and sums rounded separately, like numpy's  value += values[corner] * weight
      const double t0 = (double)v[0] * w, t1 = (double)v[1] * w, t2 = (double)v[2] * w;
      disp[0] += t0;
      disp[1] += t1;
      disp[2] += t2;
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double d = disp[a] * magnitude;
    xyz_out[i * stride + a] = (T)(p[a] + d);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Training-time augmentations of the scene reader (reference datasets/freemask_semseg.py:334-406): every geometric
// step there — centring + random shift, axis flips, volumentations' per-axis scale and rotations about z / y / x — is
// x <- x M^T + t on the first three columns of a row-major table; colours go through per-channel 256-entry tables
// (how albumentations applies RandomBrightnessContrast / RGBShift to uint8 images).  Both are one pass over the rows.
template <typename T>
__global__ __launch_bounds__(256) void affine_rows_kernel(T* __restrict__ x, int64_t n, int row_stride, double m00,
                                                          double m01, double m02, double m10, double m11, double m12,
                                                          double m20, double m21, double m22, double t0, double t1,
                                                          double t2) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  T* r = x + i * row_stride;
  const double a = (double)r[0], b = (double)r[1], c = (double)r[2];
  r[0] = (T)(m00 * a + m01 * b + m02 * c + t0);
  r[1] = (T)(m10 * a + m11 * b + m12 * c + t1);
  r[2] = (T)(m20 * a + m21 * b + m22 * c + t2);
}

// Column sums of the first `cols` columns in numpy's order for `a.sum(0)` / `a.mean(0)` of a C-contiguous f32 table:
// one f32 accumulator per column, rows added first to last (numpy reduces the outer axis row by row, not pairwise).
// One lane per column walks all rows — ~1 ms for 250 k rows, off the training step's critical path — so that the
// centred coordinates, and with them the voxel every point falls into, equal the reference's bit for bit.
__global__ __launch_bounds__(64) void colsum_sequential_kernel(const float* __restrict__ x, int64_t n, int row_stride,
                                                               int cols, float* __restrict__ out) {
  const int c = threadIdx.x;
  if (c >= cols) return;
  float acc = 0.f;
  int64_t i = 0;
  for (; i + 8 <= n; i += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = x[(i + u) * row_stride + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; i < n; ++i) acc += x[i * row_stride + c];
  out[c] = acc;
}

// out[i,c] = lut[c][ (uint8) color[i,c] ]  (the reference truncates its float colours with .astype(np.uint8) first)
__global__ __launch_bounds__(256) void color_lut_kernel(const float* __restrict__ color, int64_t n, int row_stride,
                                                        const float* __restrict__ lut, float* __restrict__ out,
                                                        int out_stride) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = color[i * row_stride + c];
    const int q = (int)(unsigned char)(int)v;            // C-style truncation + wrap, like numpy's astype(uint8)
    out[i * out_stride + c] = lut[c * 256 + q];
  }
}

// occupies `wgs` workgroups for `ticks` of the 100 MHz constant clock (usc_spin: stream placement probe)
__global__ void spin_kernel(long ticks) {
  long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
}

}  // namespace usc

using namespace usc;

extern "C" {

int usc_spin(int64_t microseconds, int wgs, usc_stream_t s) {
  USC_REQUIRE(microseconds >= 0 && microseconds <= 100000 && wgs >= 1 && wgs <= 4096, "usc_spin: bad arguments");
  hipLaunchKernelGGL(spin_kernel, dim3((unsigned)wgs), dim3(64), 0, as_stream(s), (long)(100 * microseconds));
  USC_CHECK_LAUNCH("usc_spin");
  return USC_OK;
}

int usc_knn1(const float* query, int64_t nq, const float* ref, int64_t nr, int64_t* idx, float* dist2, usc_stream_t s) {
  USC_REQUIRE(nq >= 0 && nr >= 1, "usc_knn1: bad sizes");
  if (nq == 0) return USC_OK;
  USC_REQUIRE(query && ref && idx, "usc_knn1: null pointer");
  hipLaunchKernelGGL(knn1_kernel, dim3((unsigned)ceil_div(nq, 256)), dim3(256), 0, as_stream(s), query, nq, ref, nr, idx,
                     dist2);
  USC_CHECK_LAUNCH("usc_knn1");
  return USC_OK;
}

// The propagation is iterated by the CALLER (unscene3d_amd/ops.py: cc_eps): the library never
// synchronises, the host reads the 4-byte `changed` flag between rounds.
int usc_cc_eps_init(int32_t* label, int64_t n, usc_stream_t s) {
  USC_REQUIRE(n >= 0, "usc_cc_eps_init: bad n");
  if (n == 0) return USC_OK;
  USC_REQUIRE(label, "usc_cc_eps_init: null pointer");
  hipLaunchKernelGGL(cc_init_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(s), label, n);
  USC_CHECK_LAUNCH("usc_cc_eps_init");
  return USC_OK;
}

int usc_cc_eps_step(const float* xyz, int64_t n, float eps, const int32_t* label_in, int32_t* label_out,
                    int32_t* changed, usc_stream_t s) {
  USC_REQUIRE(n >= 0 && eps >= 0.f, "usc_cc_eps_step: bad argument");
  if (n == 0) return USC_OK;
  USC_REQUIRE(xyz && label_in && label_out && changed, "usc_cc_eps_step: null pointer");
  hipStream_t st = as_stream(s);
  const unsigned g = (unsigned)ceil_div(n, 256);
  (void)hipMemsetAsync(changed, 0, 4, st);
  hipLaunchKernelGGL(cc_step_kernel, dim3(g), dim3(256), 0, st, xyz, n, (double)eps * (double)eps, label_in, label_out,
                     changed);
  hipLaunchKernelGGL(cc_jump_kernel, dim3(g), dim3(256), 0, st, label_out, n);
  USC_CHECK_LAUNCH("usc_cc_eps_step");
  return USC_OK;
}

int usc_cc_eps_finish(const int32_t* label, int64_t n, int32_t* rank_ws, int64_t* labels, usc_stream_t s) {
  USC_REQUIRE(n >= 0, "usc_cc_eps_finish: bad n");
  if (n == 0) return USC_OK;
  USC_REQUIRE(label && rank_ws && labels, "usc_cc_eps_finish: null pointer");
  hipStream_t st = as_stream(s);
  const unsigned g = (unsigned)ceil_div(n, 256);
  hipLaunchKernelGGL(cc_rootflag_kernel, dim3(g), dim3(256), 0, st, label, n, rank_ws);
  hipLaunchKernelGGL(cc_scan_kernel, dim3(1), dim3(1024), 0, st, rank_ws, n);
  hipLaunchKernelGGL(cc_relabel_kernel, dim3(g), dim3(256), 0, st, label, (const int32_t*)rank_ws, n, labels);
  USC_CHECK_LAUNCH("usc_cc_eps_finish");
  return USC_OK;
}

int usc_project_planes_fwd(const int32_t* coords, const float* pred, const float* target, int64_t V, int32_t inst,
                           int32_t dim_x, int32_t dim_y, int32_t dim_z, float* pred_xy, float* pred_xz, float* pred_yz,
                           float* tgt_xy, float* tgt_xz, float* tgt_yz, int32_t* cnt_xy, int32_t* cnt_xz, int32_t* cnt_yz,
                           usc_stream_t s) {
  USC_REQUIRE(V >= 0 && inst >= 1, "usc_project_planes_fwd: bad sizes");
  if (V == 0) return USC_OK;
  USC_REQUIRE(coords && pred && target && pred_xy && pred_xz && pred_yz && tgt_xy && tgt_xz && tgt_yz && cnt_xy &&
                  cnt_xz && cnt_yz, "usc_project_planes_fwd: null pointer");
  hipLaunchKernelGGL(project_fwd_kernel, dim3((unsigned)ceil_div(V, 256)), dim3(256), 0, as_stream(s), coords, pred, target,
                     V, (int)inst, (int)dim_x, (int)dim_y, (int)dim_z, pred_xy, pred_xz, pred_yz, tgt_xy, tgt_xz, tgt_yz,
                     cnt_xy, cnt_xz, cnt_yz);
  USC_CHECK_LAUNCH("usc_project_planes_fwd");
  return USC_OK;
}

int usc_project_planes_bwd(const int32_t* coords, int64_t V, int32_t inst, int32_t dim_x, int32_t dim_y, int32_t dim_z,
                           const float* g_xy, const float* g_xz, const float* g_yz, float* grad_pred, usc_stream_t s) {
  USC_REQUIRE(V >= 0 && inst >= 1, "usc_project_planes_bwd: bad sizes");
  if (V == 0) return USC_OK;
  USC_REQUIRE(coords && g_xy && g_xz && g_yz && grad_pred, "usc_project_planes_bwd: null pointer");
  hipLaunchKernelGGL(project_bwd_kernel, dim3((unsigned)ceil_div(V, 256)), dim3(256), 0, as_stream(s), coords, V, (int)inst,
                     (int)dim_x, (int)dim_y, (int)dim_z, g_xy, g_xz, g_yz, grad_pred);
  USC_CHECK_LAUNCH("usc_project_planes_bwd");
  return USC_OK;
}

int usc_elastic_displace(const void* xyz_in, int32_t is_f64, int64_t n, int32_t row_stride, const float* noise,
                         int32_t dim_x, int32_t dim_y, int32_t dim_z, const double* axis_x, const double* axis_y,
                         const double* axis_z, double magnitude, void* xyz_out, usc_stream_t s) {
  USC_REQUIRE(n >= 0 && row_stride >= 3 && dim_x >= 2 && dim_y >= 2 && dim_z >= 2, "usc_elastic_displace: bad sizes");
  if (n == 0) return USC_OK;
  USC_REQUIRE(xyz_in && noise && axis_x && axis_y && axis_z && xyz_out, "usc_elastic_displace: null pointer");
  const dim3 grid((unsigned)ceil_div(n, 256));
  if (is_f64)
    hipLaunchKernelGGL(elastic_displace_kernel<double>, grid, dim3(256), 0, as_stream(s), (const double*)xyz_in, n,
                       row_stride, noise, dim_x, dim_y, dim_z, axis_x, axis_y, axis_z, magnitude, (double*)xyz_out);
  else
    hipLaunchKernelGGL(elastic_displace_kernel<float>, grid, dim3(256), 0, as_stream(s), (const float*)xyz_in, n,
                       row_stride, noise, dim_x, dim_y, dim_z, axis_x, axis_y, axis_z, magnitude, (float*)xyz_out);
  USC_CHECK_LAUNCH("usc_elastic_displace");
  return USC_OK;
}

int usc_affine_rows(void* x, int32_t is_f64, int64_t n, int32_t row_stride, const double* M, const double* t,
                    usc_stream_t s) {
  USC_REQUIRE(n >= 0 && row_stride >= 3, "usc_affine_rows: bad sizes");
  if (n == 0) return USC_OK;
  USC_REQUIRE(x && M && t, "usc_affine_rows: null pointer (M, t are HOST arrays of 9 and 3 doubles)");
  const dim3 grid((unsigned)ceil_div(n, 256));
  if (is_f64)
    hipLaunchKernelGGL(affine_rows_kernel<double>, grid, dim3(256), 0, as_stream(s), (double*)x, n, (int)row_stride, M[0],
                       M[1], M[2], M[3], M[4], M[5], M[6], M[7], M[8], t[0], t[1], t[2]);
  else
    hipLaunchKernelGGL(affine_rows_kernel<float>, grid, dim3(256), 0, as_stream(s), (float*)x, n, (int)row_stride, M[0],
                       M[1], M[2], M[3], M[4], M[5], M[6], M[7], M[8], t[0], t[1], t[2]);
  USC_CHECK_LAUNCH("usc_affine_rows");
  return USC_OK;
}

int usc_colsum_sequential(const float* x, int64_t n, int32_t row_stride, int32_t cols, float* out, usc_stream_t s) {
  USC_REQUIRE(n >= 0 && cols >= 1 && cols <= 64 && row_stride >= cols, "usc_colsum_sequential: bad sizes");
  USC_REQUIRE(x && out, "usc_colsum_sequential: null pointer");
  hipLaunchKernelGGL(colsum_sequential_kernel, dim3(1), dim3(64), 0, as_stream(s), x, n, (int)row_stride, (int)cols, out);
  USC_CHECK_LAUNCH("usc_colsum_sequential");
  return USC_OK;
}

int usc_color_lut(const float* color, int64_t n, int32_t row_stride, const float* lut, float* out, int32_t out_stride,
                  usc_stream_t s) {
  USC_REQUIRE(n >= 0 && row_stride >= 3 && out_stride >= 3, "usc_color_lut: bad sizes");
  if (n == 0) return USC_OK;
  USC_REQUIRE(color && lut && out, "usc_color_lut: null pointer");
  hipLaunchKernelGGL(color_lut_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(s), color, n,
                     (int)row_stride, lut, out, (int)out_stride);
  USC_CHECK_LAUNCH("usc_color_lut");
  return USC_OK;
}

}  // extern "C"
