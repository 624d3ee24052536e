// host_coords.cpp — HIP-FREE voxelisation entry points of libusc3d_hip.so.
//
// The reference voxelises in FORKED DataLoader workers on the CPU (datasets/utils.py:403-414 calls
// ME.utils.sparse_quantize from FreeMaskVoxelizeCollate; conf/data/indoor.yaml:24 num_workers = 4): a maintainer who
// keeps that DataLoader needs a `sparse_quantize` that runs in a worker process.  A forked child must not touch the
// HIP runtime the parent initialised, so this translation unit includes NO HIP header, makes no HIP call and keeps no
// mutable global state (the error string is the library's usual thread-local channel): plain C++ over host pointers,
// safe after fork() and from several threads at once.
//
// Results are bit-equal to the device path (usc_voxel_floor_f64 + usc_coordmap_build with quant = 1): the same IEEE
// f64 division + floor, the same first-occurrence rule, rows in ascending order of first occurrence.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/usc3d.h"

namespace usc {
void set_error(const char* fmt, ...);   // coords.hip (plain host code)
}

namespace {

constexpr int kBits = 18;
constexpr int kBias = 1 << (kBits - 1);
constexpr uint64_t kEmpty = ~0ull;

inline bool in_range(int b, int x, int y, int z) {
  return b >= 0 && b < 1024 && x >= -kBias && x < kBias && y >= -kBias && y < kBias && z >= -kBias && z < kBias;
}
inline uint64_t pack(int b, int x, int y, int z) {      // the device maps' key (common.h: pack_key)
  return ((uint64_t)(uint32_t)b << (3 * kBits)) | ((uint64_t)(uint32_t)(x + kBias) << (2 * kBits)) |
         ((uint64_t)(uint32_t)(y + kBias) << kBits) | (uint64_t)(uint32_t)(z + kBias);
}
inline uint64_t mix(uint64_t k) {
  k ^= k >> 30; k *= 0xbf58476d1ce4e5b9ull;
  k ^= k >> 27; k *= 0x94d049bb133111ebull;
  k ^= k >> 31;
  return k;
}

}  // namespace

extern "C" {

int usc_voxel_floor_f64_host(const double* xyz, int64_t n, double voxel_size, int32_t* coords_out) {
  if (n < 0 || !(voxel_size > 0)) { usc::set_error("usc_voxel_floor_f64_host: bad n/voxel_size"); return USC_ERR_ARG; }
  if (n == 0) return USC_OK;
  if (!xyz || !coords_out) { usc::set_error("usc_voxel_floor_f64_host: null pointer"); return USC_ERR_ARG; }
  for (int64_t i = 0; i < 3 * n; ++i) coords_out[i] = (int32_t)floor(xyz[i] / voxel_size);    // == np.floor(x / v)
  return USC_OK;
}

int usc_unique_coords_host(const int32_t* coords, int64_t n, int32_t d, int64_t* unique_idx, int64_t* inverse,
                           int64_t* n_out) {
  if (n < 0 || (d != 3 && d != 4) || !n_out) { usc::set_error("usc_unique_coords_host: coords must be i32[n,3] or [n,4]"); return USC_ERR_ARG; }
  *n_out = 0;
  if (n == 0) return USC_OK;
  if (!coords || !unique_idx || !inverse) { usc::set_error("usc_unique_coords_host: null pointer"); return USC_ERR_ARG; }
  int64_t cap = 1024;
  while (cap < 2 * n) cap <<= 1;
  uint64_t* keys = (uint64_t*)malloc((size_t)cap * sizeof(uint64_t));
  int64_t* vals = (int64_t*)malloc((size_t)cap * sizeof(int64_t));
  if (!keys || !vals) { free(keys); free(vals); usc::set_error("usc_unique_coords_host: out of memory"); return USC_ERR_ARG; }
  memset(keys, 0xff, (size_t)cap * sizeof(uint64_t));
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int32_t* c = coords + (int64_t)d * i;
    const int b = d == 4 ? c[0] : 0, x = c[d - 3], y = c[d - 2], z = c[d - 1];
    if (!in_range(b, x, y, z)) {
      free(keys); free(vals);
      usc::set_error("usc_unique_coords_host: coordinate outside the packable range (|c| < 2^17, batch < 1024)");
      return USC_ERR_ARG;
    }
    const uint64_t key = pack(b, x, y, z);
    uint64_t slot = mix(key) & (uint64_t)(cap - 1);
    for (;;) {
      if (keys[slot] == key) { inverse[i] = vals[slot]; break; }
      if (keys[slot] == kEmpty) {          // first occurrence: rows come out in ascending order of first occurrence
        keys[slot] = key;
        vals[slot] = m;
        unique_idx[m] = i;
        inverse[i] = m;
        ++m;
        break;
      }
      slot = (slot + 1) & (uint64_t)(cap - 1);
    }
  }
  free(keys);
  free(vals);
  *n_out = m;
  return USC_OK;
}

}  // extern "C"
