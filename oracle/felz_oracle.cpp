// oracle/felz_oracle.cpp — CPU restatement of the reference's Felzenszwalb mesh over-segmentation
// (RozDavid/UnScene3D utils/cpp_utils/segmentator.cpp:17-154 `segment_graph` / `segment_mesh`, :156-250 the numpy
// wrapper's relabelling; disjoint-set forest of utils/cpp_utils/include/segmentator.h:45-93).
//
// TEST INFRASTRUCTURE ONLY: used by tests/, by tests/golden/make_golden.py and by tools/felz_bench.py as the checker /
// CPU baseline.  Nothing under unscene3d_amd/ links or loads it.
//
// Pinned: tests/test_felzenszwalb.py::test_oracle_equals_reference_build compares it with the reference's own
// extension module built from /root/reference (oracle/Makefile target `ref`, output oracle/_ref/) on seeded meshes —
// labels and connectivity identical — and the committed fixtures tests/golden/felz.npz were produced by that build.
//
// Arithmetic notes that matter for bit-equality with the reference (compiled for baseline x86-64: no FMA):
//   * face normal = normalised cross product, blended into the three vertex normals by a running lerp in FACE ORDER
//     with weight 1/(count+1) (segmentator.cpp:62-82): order dependent, so restated as the same loop;
//   * edge weight = (1 - n_a.n_b) * sum|c_a - c_b|, squared when the edge leaves b's tangent plane on the convex side
//     and the colours are close (`color_dist < 0.05` is a DOUBLE comparison) (:85-121);
//   * `std::sort` on the weights (unstable: the order of equal weights is libstdc++'s; `stable` != 0 switches to
//     std::stable_sort, the canonical order of the device path) (:18);
//   * threshold rule w <= thr[a] && w <= thr[b], thr = w + c/size after a join (:27-42); small-segment joins in
//     sorted edge order (:127-133); output = representative per vertex (:136-139).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace {

struct Edge { float w; int a, b; };
inline bool operator<(const Edge& x, const Edge& y) { return x.w < y.w; }

struct Forest {
  struct Elt { int rank, p, size; };
  std::vector<Elt> e;
  explicit Forest(int n) : e(n) { for (int i = 0; i < n; ++i) e[i] = Elt{0, i, 1}; }
  int find(int x) {
    int y = x;
    while (y != e[y].p) y = e[y].p;
    e[x].p = y;                       // "path compression (sort of)": only x is re-pointed
    return y;
  }
  void join(int x, int y) {
    if (e[x].rank > e[y].rank) { e[y].p = x; e[x].size += e[y].size; }
    else { e[x].p = y; e[y].size += e[x].size; if (e[x].rank == e[y].rank) e[y].rank++; }
  }
  int size(int x) const { return e[x].size; }
};

struct V3 { float x, y, z; };
inline V3 sub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 unit_cross(V3 u, V3 v) {
  V3 c{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
  const float n = sqrtf(c.x * c.x + c.y * c.y + c.z * c.z);
  c.x /= n; c.y /= n; c.z /= n;
  return c;
}
inline V3 lerp(V3 a, V3 b, float v) {
  const float u = 1.0f - v;
  return V3{v * b.x + u * a.x, v * b.y + u * a.y, v * b.z + u * a.z};
}

}  // namespace

extern "C" {

// vertices, colors f32[nv,3]; faces i32[nf,3].  Outputs: comps i32[nv] (union-find representative per vertex),
// optionally normals f32[nv,3], weights f32[3 nf] (edge order: face-major, (i1,i2),(i1,i3),(i3,i2)),
// sorted_a/sorted_b i32[3 nf] (the edge order the merge loops used).  Any output pointer may be NULL except comps.
int felz_oracle_segment(const float* vertices, const float* colors, const int32_t* faces, int32_t nv, int32_t nf,
                        float kthr, int32_t seg_min_verts, int32_t stable, int32_t* comps, float* normals_out,
                        float* weights_out, int32_t* sorted_a, int32_t* sorted_b) {
  const V3* P = reinterpret_cast<const V3*>(vertices);
  const V3* C = reinterpret_cast<const V3*>(colors);
  const int64_t ne = (int64_t)nf * 3;
  std::vector<Edge> edges(ne);
  std::vector<V3> N(nv, V3{0.f, 0.f, 0.f});
  std::vector<int> counts(nv, 0);
  for (int i = 0; i < nf; ++i) {
    const int i1 = faces[3 * i], i2 = faces[3 * i + 1], i3 = faces[3 * i + 2];
    const int eb = 3 * i;
    edges[eb].a = i1; edges[eb].b = i2;
    edges[eb + 1].a = i1; edges[eb + 1].b = i3;
    edges[eb + 2].a = i3; edges[eb + 2].b = i2;
    const V3 n = unit_cross(sub(P[i2], P[i1]), sub(P[i3], P[i1]));
    N[i1] = lerp(N[i1], n, 1.0f / (counts[i1] + 1.0f));
    N[i2] = lerp(N[i2], n, 1.0f / (counts[i2] + 1.0f));
    N[i3] = lerp(N[i3], n, 1.0f / (counts[i3] + 1.0f));
    counts[i1]++; counts[i2]++; counts[i3]++;
  }
  for (int64_t i = 0; i < ne; ++i) {
    const int a = edges[i].a, b = edges[i].b;
    const V3 n1 = N[a], n2 = N[b], p1 = P[a], p2 = P[b];
    float dx = p2.x - p1.x, dy = p2.y - p1.y, dz = p2.z - p1.z;
    const float dd = sqrtf(dx * dx + dy * dy + dz * dz);
    dx /= dd; dy /= dd; dz /= dd;
    const float dot = n1.x * n2.x + n1.y * n2.y + n1.z * n2.z;
    const float normal_dist = 1.0f - dot;
    const float color_dist = fabsf(C[a].x - C[b].x) + fabsf(C[a].y - C[b].y) + fabsf(C[a].z - C[b].z);
    float dist = normal_dist * color_dist;
    const float dot2 = n2.x * dx + n2.y * dy + n2.z * dz;
    if (dot2 > 0 && color_dist < 0.05) dist = dist * dist;
    edges[i].w = dist;
    if (weights_out) weights_out[i] = dist;
  }
  if (normals_out)
    for (int i = 0; i < nv; ++i) { normals_out[3 * i] = N[i].x; normals_out[3 * i + 1] = N[i].y; normals_out[3 * i + 2] = N[i].z; }
  if (stable) std::stable_sort(edges.begin(), edges.end());
  else std::sort(edges.begin(), edges.end());
  Forest u(nv);
  std::vector<float> thr(nv, kthr);
  for (int64_t i = 0; i < ne; ++i) {
    int a = u.find(edges[i].a);
    const int b = u.find(edges[i].b);
    if (a != b && edges[i].w <= thr[a] && edges[i].w <= thr[b]) {
      u.join(a, b);
      a = u.find(a);
      thr[a] = edges[i].w + (kthr / u.size(a));
    }
  }
  for (int64_t j = 0; j < ne; ++j) {
    const int a = u.find(edges[j].a), b = u.find(edges[j].b);
    if (a != b && (u.size(a) < seg_min_verts || u.size(b) < seg_min_verts)) u.join(a, b);
  }
  for (int q = 0; q < nv; ++q) comps[q] = u.find(q);
  for (int64_t i = 0; i < ne; ++i) {
    if (sorted_a) sorted_a[i] = edges[i].a;
    if (sorted_b) sorted_b[i] = edges[i].b;
  }
  return 0;
}

}  // extern "C"
