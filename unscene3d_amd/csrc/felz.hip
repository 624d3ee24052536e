// felz.hip — Felzenszwalb mesh over-segmentation: the producer of `segment_ids` /
// `seg_connectivity` for the pseudo-mask generator and the self-training targets.
// Replaces the reference's felzenszwalb_cpp extension (utils/cpp_utils/segmentator.cpp:17-154; caller
// pseudo_masks/datasets/scannet.py:156-197).
//
// Split of the work:
//   device  face normals; vertex normals = the reference's running blend of face normals IN FACE ORDER, evaluated per
//           vertex over its incident faces (a stable vertex->corner CSR keeps the order); edge weights.  Every
//           multiply/add is rounded separately (fp contract off) and sqrt / division are IEEE, so normals and weights
//           equal the reference's bit for bit (it is compiled for baseline x86-64: no FMA).
//   device  the sort of the 3F edge weights (caller: a stable device sort; equal weights keep face order — the
//           reference's std::sort leaves their order to libstdc++, which changes the union-find representatives but
//           not the partition, see tests/test_felzenszwalb.py)
//   host    the two merge loops (usc_felz_merge_host): every join changes the threshold the next edge is tested
//           against, so they are one sequential chain over the sorted edges (~10 ns per edge on a CPU core; a single
//           GPU lane would need ~1 us per dependent global access).
#include <math.h>

#include <vector>

#include "common.h"

#pragma clang fp contract(off)

namespace usc {
namespace {

struct V3 { float x, y, z; };
__device__ inline V3 ld3(const float* p, int64_t i) { return V3{p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }

__global__ __launch_bounds__(256) void felz_face_normals_kernel(const float* __restrict__ P, const int32_t* __restrict__ faces,
                                                                int64_t nf, float* __restrict__ fn) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nf) return;
  const V3 p1 = ld3(P, faces[3 * f]), p2 = ld3(P, faces[3 * f + 1]), p3 = ld3(P, faces[3 * f + 2]);
  const V3 u{p2.x - p1.x, p2.y - p1.y, p2.z - p1.z}, v{p3.x - p1.x, p3.y - p1.y, p3.z - p1.z};
  V3 c{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};       // segmentator.h:99-104
  const float n = sqrtf(c.x * c.x + c.y * c.y + c.z * c.z);
  fn[3 * f] = c.x / n; fn[3 * f + 1] = c.y / n; fn[3 * f + 2] = c.z / n;
}

// order: the 3F (face, corner) entries sorted by vertex, stable (so each vertex sees its faces in face order);
// off[v] .. off[v+1] delimit vertex v.  normal <- lerp(normal, face normal, 1/(count+1)), count = faces seen before
// the current one (segmentator.cpp:62-82: the three corners of a face use the counts from before the face).
__global__ __launch_bounds__(256) void felz_vertex_normals_kernel(const float* __restrict__ fn, const int64_t* __restrict__ order,
                                                                  const int64_t* __restrict__ off, int64_t nv,
                                                                  float* __restrict__ normals) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  V3 n{0.f, 0.f, 0.f};
  int cnt = 0, pend = 0;
  int64_t cur = -1;
  for (int64_t q = off[v]; q < off[v + 1]; ++q) {
    const int64_t f = order[q] / 3;
    if (f != cur) { cnt += pend; pend = 0; cur = f; }
    const float t = 1.0f / ((float)cnt + 1.0f);
    const float u = 1.0f - t;
    const V3 b = ld3(fn, f);
    n = V3{t * b.x + u * n.x, t * b.y + u * n.y, t * b.z + u * n.z};            // lerp, segmentator.h:105-108
    ++pend;
  }
  normals[3 * v] = n.x; normals[3 * v + 1] = n.y; normals[3 * v + 2] = n.z;
}

// edge 3f+0 = (i1,i2), 3f+1 = (i1,i3), 3f+2 = (i3,i2)   (segmentator.cpp:72-74, weights :85-121)
__global__ __launch_bounds__(256) void felz_edge_weights_kernel(const float* __restrict__ P, const float* __restrict__ C,
                                                                const float* __restrict__ N, const int32_t* __restrict__ faces,
                                                                int64_t ne, int32_t* __restrict__ ea, int32_t* __restrict__ eb,
                                                                float* __restrict__ w) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  const int64_t f = e / 3;
  const int j = (int)(e - 3 * f);
  const int32_t i1 = faces[3 * f], i2 = faces[3 * f + 1], i3 = faces[3 * f + 2];
  const int32_t a = j == 2 ? i3 : i1, b = j == 1 ? i3 : i2;
  const V3 n1 = ld3(N, a), n2 = ld3(N, b), p1 = ld3(P, a), p2 = ld3(P, b), c1 = ld3(C, a), c2 = ld3(C, b);
  float dx = p2.x - p1.x, dy = p2.y - p1.y, dz = p2.z - p1.z;
  const float dd = sqrtf(dx * dx + dy * dy + dz * dz);
  dx /= dd; dy /= dd; dz /= dd;
  const float dot = n1.x * n2.x + n1.y * n2.y + n1.z * n2.z;
  const float normal_dist = 1.0f - dot;
  const float color_dist = fabsf(c1.x - c2.x) + fabsf(c1.y - c2.y) + fabsf(c1.z - c2.z);
  float dist = normal_dist * color_dist;
  const float dot2 = n2.x * dx + n2.y * dy + n2.z * dz;
  if (dot2 > 0 && (double)color_dist < 0.05) dist = dist * dist;     // the reference compares against a double literal
  ea[e] = a; eb[e] = b; w[e] = dist;
}

}  // namespace
}  // namespace usc

using namespace usc;

extern "C" {

int usc_felz_face_normals(const float* vertices, const int32_t* faces, int64_t n_faces, float* face_normals,
                          usc_stream_t s) {
  USC_REQUIRE(n_faces >= 0, "usc_felz_face_normals: bad size");
  if (n_faces == 0) return USC_OK;
  USC_REQUIRE(vertices && faces && face_normals, "usc_felz_face_normals: null pointer");
  hipLaunchKernelGGL(felz_face_normals_kernel, dim3((unsigned)ceil_div(n_faces, 256)), dim3(256), 0, as_stream(s), vertices,
                     faces, n_faces, face_normals);
  USC_CHECK_LAUNCH("usc_felz_face_normals");
  return USC_OK;
}

int usc_felz_vertex_normals(const float* face_normals, const int64_t* corner_order, const int64_t* vertex_off,
                            int64_t n_vertices, float* normals, usc_stream_t s) {
  USC_REQUIRE(n_vertices >= 0, "usc_felz_vertex_normals: bad size");
  if (n_vertices == 0) return USC_OK;
  USC_REQUIRE(face_normals && corner_order && vertex_off && normals, "usc_felz_vertex_normals: null pointer");
  hipLaunchKernelGGL(felz_vertex_normals_kernel, dim3((unsigned)ceil_div(n_vertices, 256)), dim3(256), 0, as_stream(s),
                     face_normals, corner_order, vertex_off, n_vertices, normals);
  USC_CHECK_LAUNCH("usc_felz_vertex_normals");
  return USC_OK;
}

int usc_felz_edge_weights(const float* vertices, const float* colors, const float* normals, const int32_t* faces,
                          int64_t n_faces, int32_t* edge_a, int32_t* edge_b, float* weights, usc_stream_t s) {
  USC_REQUIRE(n_faces >= 0, "usc_felz_edge_weights: bad size");
  if (n_faces == 0) return USC_OK;
  USC_REQUIRE(vertices && colors && normals && faces && edge_a && edge_b && weights, "usc_felz_edge_weights: null pointer");
  const int64_t ne = 3 * n_faces;
  hipLaunchKernelGGL(felz_edge_weights_kernel, dim3((unsigned)ceil_div(ne, 256)), dim3(256), 0, as_stream(s), vertices,
                     colors, normals, faces, ne, edge_a, edge_b, weights);
  USC_CHECK_LAUNCH("usc_felz_edge_weights");
  return USC_OK;
}

// HOST pointers (the only entry point of this library that takes them): the sequential part.
int usc_felz_merge_host(const int32_t* edge_a, const int32_t* edge_b, const float* weights, int64_t n_edges,
                        int32_t n_vertices, float kthr, int32_t seg_min_verts, int32_t* comps) {
  USC_REQUIRE(n_edges >= 0 && n_vertices >= 0, "usc_felz_merge_host: bad sizes");
  USC_REQUIRE((n_edges == 0 || (edge_a && edge_b && weights)) && (n_vertices == 0 || comps),
              "usc_felz_merge_host: null pointer");
  for (int64_t i = 0; i < n_edges; ++i)
    USC_REQUIRE(edge_a[i] >= 0 && edge_a[i] < n_vertices && edge_b[i] >= 0 && edge_b[i] < n_vertices,
                "usc_felz_merge_host: edge %lld references a vertex outside [0, %d)", (long long)i, n_vertices);
  // disjoint-set forest with union by rank; find re-points only the queried element (segmentator.h:45-93)
  std::vector<int32_t> parent(n_vertices), rank(n_vertices, 0), size(n_vertices, 1);
  std::vector<float> thr(n_vertices, kthr);
  for (int32_t i = 0; i < n_vertices; ++i) parent[i] = i;
  auto find = [&](int32_t x) {
    int32_t y = x;
    while (y != parent[y]) y = parent[y];
    parent[x] = y;
    return y;
  };
  auto join = [&](int32_t x, int32_t y) {
    if (rank[x] > rank[y]) { parent[y] = x; size[x] += size[y]; }
    else { parent[x] = y; size[y] += size[x]; if (rank[x] == rank[y]) ++rank[y]; }
  };
  for (int64_t i = 0; i < n_edges; ++i) {            // segment_graph: non-decreasing weight order (segmentator.cpp:26-42)
    int32_t a = find(edge_a[i]);
    const int32_t b = find(edge_b[i]);
    const float w = weights[i];
    if (a != b && w <= thr[a] && w <= thr[b]) {
      join(a, b);
      a = find(a);
      thr[a] = w + (kthr / (float)size[a]);
    }
  }
  for (int64_t j = 0; j < n_edges; ++j) {            // small segments join a neighbour, same edge order (:127-133)
    const int32_t a = find(edge_a[j]), b = find(edge_b[j]);
    if (a != b && (size[a] < seg_min_verts || size[b] < seg_min_verts)) join(a, b);
  }
  for (int32_t q = 0; q < n_vertices; ++q) comps[q] = find(q);
  return USC_OK;
}

}  // extern "C"
