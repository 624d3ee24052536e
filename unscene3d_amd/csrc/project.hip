// project.hip — 2D -> 3D feature projection:
//   first-hit ray cast of every pixel through the voxel map   (project_image_cuda_kernel.cu:24-56,113-146)
//   per-voxel reduction of the hit pixels' features            (project_image_cuda_kernel.cu:58-65, raycast_image.py:66-68)
//   running-mean fusion of a frame into the scene features     (pseudo_masks/unscene3d_pseudo_main.py:311-330)
//   prediction mode (integer max) and depth-image unprojection (project_image_cuda_kernel.cu:68-109,249-290)
//
// The reference marches every ray in steps of voxel_size/2 *voxel units* (0.01 voxel: ~19 500 samples per ray) and
// reads a dense int64 occupancy grid at every sample, then adds the pixel's feature vector to its voxel with one float
// atomic per channel.  Here:
//   * the voxel lookup goes to the coordinate hash the convolutions already built (no dense grid); a dense-grid
//     variant exists for callers that hand over the reference's occupancy tensor;
//   * the march keeps the reference's sample positions bit for bit (same sequential fp32 accumulation of the ray
//     parameter) but evaluates the voxel only every kJump samples: every coordinate of a sample is a monotone function
//     of the ray parameter, so two samples in the same voxel pin every sample between them to that voxel;
//   * the features are reduced per voxel over a CSR of the hit pixels (stable, ascending pixel order): deterministic,
//     coalesced row reads, no atomics, and only the rows that were hit are touched.
#include "common.h"

namespace usc {

struct Ray {
  float cx, cy, cz;   // camera position
  float dx, dy, dz;   // unit direction
  float t0, t1;       // ray parameter range
};

// Every multiply and add rounds separately (the oracle is numpy fp32); 1/sqrt and divisions are IEEE.
__device__ inline Ray make_ray(const float* __restrict__ m, const float* __restrict__ k, int x, int y, float dmin,
                               float dmax) {
#pragma clang fp contract(off)
  const float fx = k[0], fy = k[1], mx = k[2], my = k[3];
  const float depth = 1.0f * (dmax - dmin) + dmin;
  float ax = depth * (((float)x - mx) / fx), ay = depth * (((float)y - my) / fy), az = depth;
  float inv = 1.0f / sqrtf(ax * ax + ay * ay + az * az);
  ax = ax * inv; ay = ay * inv; az = az * inv;
  float wx = m[0] * ax + m[1] * ay + m[2] * az + m[3] * 0.0f;
  float wy = m[4] * ax + m[5] * ay + m[6] * az + m[7] * 0.0f;
  float wz = m[8] * ax + m[9] * ay + m[10] * az + m[11] * 0.0f;
  inv = 1.0f / sqrtf(wx * wx + wy * wy + wz * wz);
  Ray r;
  r.cx = m[3]; r.cy = m[7]; r.cz = m[11];
  r.dx = wx * inv; r.dy = wy * inv; r.dz = wz * inv;
  const float to_len = 1.0f / az;
  r.t0 = to_len * dmin;
  r.t1 = to_len * dmax;
  return r;
}

__device__ inline int round_away(float p) {
#pragma clang fp contract(off)
  const float s = p > 0.0f ? 0.5f : (p < 0.0f ? -0.5f : 0.0f);
  return (int)(p + s);
}

struct Vox { int x, y, z; };
__device__ inline Vox voxel_at(const Ray& r, float t) {
#pragma clang fp contract(off)
  Vox v;
  v.x = round_away(r.cx + t * r.dx);
  v.y = round_away(r.cy + t * r.dy);
  v.z = round_away(r.cz + t * r.dz);
  return v;
}
__device__ inline bool same(const Vox& a, const Vox& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

struct HashOcc {
  const uint64_t* keys;
  const int32_t* vals;
  int64_t cap;
  const int32_t* shift;   // i32[B,3]: local voxel (0-based) + shift = map coordinate
  // optional occupancy of 8^3-voxel bricks (one bit each, rows 1.. only): a ray spends most of its voxels in free
  // space, where one bit from a KB-sized, cache-resident mask replaces a probe into the MB-sized hash table
  const uint32_t* bricks;
  int bx, by, bz, brick_words;
  __device__ int operator()(int b, const Vox& v) const {
    if (v.x < 0 || v.y < 0 || v.z < 0) return 0;
    if (bricks) {
      const int gx = v.x >> 3, gy = v.y >> 3, gz = v.z >> 3;
      if (gx >= bx || gy >= by || gz >= bz) return 0;
      const int idx = (gz * by + gy) * bx + gx;
      if (!((bricks[(int64_t)b * brick_words + (idx >> 5)] >> (idx & 31)) & 1u)) return 0;
    }
    const int x = v.x + shift[b * 3 + 0], y = v.y + shift[b * 3 + 1], z = v.z + shift[b * 3 + 2];
    if (!coord_in_range(b, x, y, z)) return 0;
    const int row = table_lookup(keys, vals, cap, pack_key(b, x, y, z));
    return row > 0 ? row : 0;   // row 0 reads as "empty" in the reference's occupancy grid
  }
};
__global__ __launch_bounds__(256) void brick_mask_kernel(const int32_t* __restrict__ coords, int64_t n,
                                                        const int32_t* __restrict__ shift, int bx, int by, int bz,
                                                        int brick_words, uint32_t* __restrict__ mask) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n || i == 0) return;   // row 0 never hits
  const int b = coords[i * 4];
  const int gx = (coords[i * 4 + 1] - shift[b * 3 + 0]) >> 3, gy = (coords[i * 4 + 2] - shift[b * 3 + 1]) >> 3,
            gz = (coords[i * 4 + 3] - shift[b * 3 + 2]) >> 3;
  if (gx < 0 || gy < 0 || gz < 0 || gx >= bx || gy >= by || gz >= bz) return;
  const int idx = (gz * by + gy) * bx + gx;
  atomicOr(&mask[(int64_t)b * brick_words + (idx >> 5)], 1u << (idx & 31));
}
struct DenseOcc {
  const int64_t* occ;
  int dz, dy, dx;
  __device__ int operator()(int b, const Vox& v) const {
    if (v.x < 0 || v.y < 0 || v.z < 0 || v.x >= dx || v.y >= dy || v.z >= dz) return 0;
    return (int)occ[(((int64_t)b * dz + v.z) * dy + v.y) * dx + v.x];
  }
};

constexpr int kJump = 32;   // power of two

// One thread per pixel, one wave per 8x8 pixel tile (neighbouring rays walk the same voxels).
// "Sample k lies outside the current voxel (or past the end of the ray)" is monotone in k: the ray parameter only
// grows and a straight ray never re-enters a voxel it left.  So the march tests sample k+kJump and, when that one is
// outside, bisects for the first outside sample (re-running the adds: the parameter of sample k is only defined by
// the chain of fp32 additions that leads to it).  One occupancy lookup per voxel entered, none per sample.
template <class Occ>
__global__ __launch_bounds__(64) void raycast_first_hit_kernel(Occ occ, const float* __restrict__ views,
                                                              const float* __restrict__ intr, int V, int H, int W,
                                                              float dmin, float dmax, float inc,
                                                              int32_t* __restrict__ hit, int64_t* __restrict__ seg,
                                                              int64_t n_rows) {
  const int x = blockIdx.x * 8 + (threadIdx.x & 7), y = blockIdx.y * 8 + (threadIdx.x >> 3);
  const int bv = blockIdx.z, b = bv / V;
  if (x >= W || y >= H) return;
  const Ray r = make_ray(views + (int64_t)bv * 16, intr + b * 4, x, y, dmin, dmax);
  float t = r.t0;
  int found = 0;
  Vox last = {INT32_MIN, 0, 0};
  while (t < r.t1) {
    const Vox v = voxel_at(r, t);
    if (!same(v, last)) {
      last = v;
      found = occ(b, v);
      if (found != 0) break;
    }
    float tj = t;
#pragma unroll
    for (int s = 0; s < kJump; ++s) tj += inc;
    if (tj < r.t1 && same(voxel_at(r, tj), last)) {
      t = tj;   // the samples in between lie in `last` as well
      continue;
    }
    // sample +0 is inside, sample +kJump is outside: bisect for the last inside sample
#pragma unroll
    for (int step = kJump / 2; step >= 1; step >>= 1) {
      float tm = t;
#pragma unroll
      for (int s = 0; s < step; ++s) tm += inc;
      if (tm < r.t1 && same(voxel_at(r, tm), last)) t = tm;
    }
    t += inc;   // first sample outside `last` (or past the end: the loop condition ends the ray)
  }
  const int64_t p = ((int64_t)bv * H + y) * W + x;
  hit[p] = found != 0 ? found : -1;
  if (seg) seg[p] = found != 0 ? (int64_t)found : n_rows;   // misses go to the extra segment n_rows
}

template <class Occ>
static int launch_raycast(const char* name, Occ occ, const float* views, const float* intrinsics, int B, int V, int H,
                          int W, float dmin, float dmax, float inc, int32_t* hit, int64_t* seg, int64_t n_rows,
                          hipStream_t st) {
  dim3 grid((W + 7) / 8, (H + 7) / 8, B * V);
  hipLaunchKernelGGL(raycast_first_hit_kernel<Occ>, grid, dim3(64), 0, st, occ, views, intrinsics, V, H, W, dmin, dmax,
                     inc, hit, seg, n_rows);
  USC_CHECK_LAUNCH(name);
  return 0;
}

// One wave per voxel row; lanes stride over the channels, pixels of the row are added in CSR order.
// mode 0: out[r,:] = sum / (count + 1e-4) for every row (zeros where nothing hit), num[r] = count
// mode 1: scene[r,:] = (scene[r,:] + sum / (count + 1e-4)) / 2 on the rows that were hit, num[r] = count
// mode 2: out[r,:] += sum, num[r] += count (the raw accumulation of the reference's operator)
__global__ __launch_bounds__(256) void project_reduce_kernel(const float* __restrict__ feats, int C,
                                                            const int64_t* __restrict__ order,
                                                            const int64_t* __restrict__ seg_off, int64_t n_rows,
                                                            int mode, float* __restrict__ out,
                                                            int32_t* __restrict__ num) {
#pragma clang fp contract(off)
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const int64_t p0 = seg_off[r], p1 = seg_off[r + 1];
  const int cnt = (int)(p1 - p0);
  if (lane == 0 && num) num[r] = mode == 2 ? num[r] + cnt : cnt;
  if (cnt == 0) {
    if (mode == 0)
      for (int c = lane; c < C; c += 64) out[r * C + c] = 0.0f;
    return;
  }
  const float den = (float)cnt + 10e-5f;
  for (int c0 = 0; c0 < C; c0 += 256) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const int c = c0 + lane;
    for (int64_t p = p0; p < p1; ++p) {
      const float* row = feats + order[p] * C;
      if (c < C) a0 += row[c];
      if (c + 64 < C) a1 += row[c + 64];
      if (c + 128 < C) a2 += row[c + 128];
      if (c + 192 < C) a3 += row[c + 192];
    }
    float a[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cc = c + 64 * j;
      if (cc < C) {
        if (mode == 2) {
          out[r * C + cc] += a[j];
        } else {
          const float m = a[j] / den;
          out[r * C + cc] = mode == 0 ? m : (out[r * C + cc] + m) / 2.0f;
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void project_preds_kernel(const int32_t* __restrict__ preds, int C,
                                                           const int32_t* __restrict__ hit, int64_t n_pix,
                                                           int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pix * C) return;
  const int64_t p = i / C;
  const int c = (int)(i - p * C);
  const int row = hit[p];
  if (row > 0) atomicMax(&out[(int64_t)row * C + c], preds[i]);
}

__global__ __launch_bounds__(256) void unproject_depth_kernel(const float* __restrict__ depth,
                                                             const float* __restrict__ views,
                                                             const float* __restrict__ intr, int H, int W,
                                                             int64_t n_pix, float* __restrict__ cloud) {
#pragma clang fp contract(off)
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n_pix) return;
  const float d = depth[p];
  if (d <= 0.0f) return;
  const int v = (int)(p / ((int64_t)H * W));
  const int rem = (int)(p - (int64_t)v * H * W);
  const int y = rem / W, x = rem - y * W;
  const float* m = views + (int64_t)v * 16;
  const float* k = intr + (int64_t)v * 4;
  const float px = ((float)x - k[2]) * d / k[0], py = ((float)y - k[3]) * d / k[1];
  cloud[p * 5 + 0] = (float)v;
  cloud[p * 5 + 1] = (float)p;
  cloud[p * 5 + 2] = m[0] * px + m[1] * py + m[2] * d + m[3];
  cloud[p * 5 + 3] = m[4] * px + m[5] * py + m[6] * d + m[7];
  cloud[p * 5 + 4] = m[8] * px + m[9] * py + m[10] * d + m[11];
}

static int check_rays(const char* name, int B, int V, int H, int W, float dmin, float dmax, float inc) {
  USC_REQUIRE(B > 0 && V > 0 && H > 0 && W > 0, "%s: empty image batch (%d,%d,%d,%d)", name, B, V, H, W);
  USC_REQUIRE((int64_t)B * V <= 65535, "%s: more than 65535 views per call", name);
  USC_REQUIRE(inc > 0.0f && dmin > 0.0f && dmax > dmin, "%s: needs 0 < depth_min < depth_max and ray_increment > 0",
              name);
  // a ray parameter that stops growing (increment below half an ulp) would never terminate
  USC_REQUIRE(inc > dmax * 4.0f * 1.1920929e-7f, "%s: ray_increment %g too small against depth_max %g", name, inc, dmax);
  return 0;
}

}  // namespace usc

using namespace usc;

extern "C" {

int usc_brick_mask_build(const int32_t* coords, int64_t n, const int32_t* shift, int32_t B, int32_t bricks_x,
                         int32_t bricks_y, int32_t bricks_z, uint32_t* mask, usc_stream_t s) {
  USC_REQUIRE(B > 0 && bricks_x > 0 && bricks_y > 0 && bricks_z > 0, "usc_brick_mask_build: empty brick grid");
  const int64_t words = ceil_div((int64_t)bricks_x * bricks_y * bricks_z, 32);
  USC_REQUIRE(words < (1 << 26), "usc_brick_mask_build: brick grid too large");
  (void)hipMemsetAsync(mask, 0, (size_t)B * words * 4, as_stream(s));
  if (n > 0)
    hipLaunchKernelGGL(brick_mask_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(s), coords, n, shift,
                       bricks_x, bricks_y, bricks_z, (int)words, mask);
  USC_CHECK_LAUNCH("usc_brick_mask_build");
  return 0;
}

int usc_raycast_first_hit_map(const uint64_t* table_keys, const int32_t* table_vals, int64_t cap, int64_t n_rows,
                              const int32_t* shift, const uint32_t* brick_mask, int32_t bricks_x, int32_t bricks_y,
                              int32_t bricks_z, const float* views, const float* intrinsics, int32_t B, int32_t V,
                              int32_t H, int32_t W, float depth_min, float depth_max, float ray_increment,
                              int32_t* hit, int64_t* seg, usc_stream_t s) {
  if (int rc = check_rays("usc_raycast_first_hit_map", B, V, H, W, depth_min, depth_max, ray_increment)) return rc;
  USC_REQUIRE(cap > 0 && (cap & (cap - 1)) == 0, "usc_raycast_first_hit_map: capacity must be a power of two");
  USC_REQUIRE(!brick_mask || (bricks_x > 0 && bricks_y > 0 && bricks_z > 0), "usc_raycast_first_hit_map: brick grid");
  HashOcc occ{table_keys, table_vals, cap, shift, brick_mask, bricks_x, bricks_y, bricks_z,
              brick_mask ? (int)ceil_div((int64_t)bricks_x * bricks_y * bricks_z, 32) : 0};
  return launch_raycast("usc_raycast_first_hit_map", occ, views, intrinsics, B, V, H, W, depth_min, depth_max,
                        ray_increment, hit, seg, n_rows, as_stream(s));
}

int usc_raycast_first_hit_dense(const int64_t* occupancy, int32_t dim_z, int32_t dim_y, int32_t dim_x, int64_t n_rows,
                                const float* views, const float* intrinsics, int32_t B, int32_t V, int32_t H,
                                int32_t W, float depth_min, float depth_max, float ray_increment, int32_t* hit,
                                int64_t* seg, usc_stream_t s) {
  if (int rc = check_rays("usc_raycast_first_hit_dense", B, V, H, W, depth_min, depth_max, ray_increment)) return rc;
  USC_REQUIRE(dim_z > 0 && dim_y > 0 && dim_x > 0, "usc_raycast_first_hit_dense: empty occupancy grid");
  DenseOcc occ{occupancy, dim_z, dim_y, dim_x};
  return launch_raycast("usc_raycast_first_hit_dense", occ, views, intrinsics, B, V, H, W, depth_min, depth_max,
                        ray_increment, hit, seg, n_rows, as_stream(s));
}

int usc_project_reduce(const float* feats, int32_t c, const int64_t* order, const int64_t* seg_off, int64_t n_rows,
                       int32_t mode, float* out, int32_t* num, usc_stream_t s) {
  USC_REQUIRE(c > 0, "usc_project_reduce: no channels");
  USC_REQUIRE(mode >= 0 && mode <= 2, "usc_project_reduce: mode %d (0 mean, 1 fuse, 2 accumulate)", mode);
  if (n_rows <= 0) return 0;
  hipLaunchKernelGGL(project_reduce_kernel, dim3((unsigned)ceil_div(n_rows, 4)), dim3(256), 0, as_stream(s), feats, c,
                     order, seg_off, n_rows, mode, out, num);
  USC_CHECK_LAUNCH("usc_project_reduce");
  return 0;
}

int usc_project_predictions(const int32_t* preds, int32_t c, const int32_t* hit, int64_t n_pix, int32_t* out,
                            usc_stream_t s) {
  USC_REQUIRE(c > 0, "usc_project_predictions: no channels");
  if (n_pix <= 0) return 0;
  hipLaunchKernelGGL(project_preds_kernel, dim3((unsigned)ceil_div(n_pix * c, 256)), dim3(256), 0, as_stream(s), preds,
                     c, hit, n_pix, out);
  USC_CHECK_LAUNCH("usc_project_predictions");
  return 0;
}

int usc_unproject_depth(const float* depth, const float* views, const float* intrinsics, int32_t V, int32_t H,
                        int32_t W, float* cloud, usc_stream_t s) {
  const int64_t n = (int64_t)V * H * W;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(unproject_depth_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(s), depth, views,
                     intrinsics, H, W, n, cloud);
  USC_CHECK_LAUNCH("usc_unproject_depth");
  return 0;
}

}  // extern "C"
