// coords.hip — coordinate maps and kernel maps.
//
// MI355X design: one open-addressing hash table per coordinate map, resident in
// HBM (keys u64 + vals i32, load factor <= 0.5), built with 64-bit CAS +
// atomicMin on the row index so that the FIRST occurrence of every voxel wins
// regardless of thread scheduling; rows are then compacted in ascending order
// by a wave-shuffle prefix sum, which reproduces the sequential insert order of
// [ME] CoordinateMapCPU::insert_and_map (unique_map ascending, first occurrence)
// bit-exactly.  Kernel maps are dense neighbour tables nbr[K][N_out] (coalesced
// per-offset row reads for the output-stationary implicit GEMM) that can be
// compacted into [ME]-style per-offset (in,out) pair lists.
#include "common.h"
#include "scan.h"

#include <stdarg.h>

namespace usc {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------------------
__global__ void voxel_floor_kernel(const double* __restrict__ xyz, int64_t n3, double voxel,
                                   int32_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n3;
       i += (int64_t)gridDim.x * blockDim.x) {
    // IEEE f64 division + floor == numpy's np.floor(x / v) element-wise
    out[i] = (int32_t)floor(xyz[i] / voxel);
  }
}

__global__ void fill_i32_kernel(int32_t* p, int64_t n, int32_t v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    p[i] = v;
}

struct CoordRow { int b, x, y, z; };

__device__ inline CoordRow load_quant(const int32_t* __restrict__ coords, int64_t i, int quant) {
  const int4 c = *reinterpret_cast<const int4*>(coords + 4 * i);
  CoordRow r{c.x, c.y, c.z, c.w};
  if (quant > 1) {
    r.x = floor_quant(r.x, quant);
    r.y = floor_quant(r.y, quant);
    r.z = floor_quant(r.z, quant);
  }
  return r;
}

__global__ __launch_bounds__(256) void coordmap_insert_kernel(
    const int32_t* __restrict__ coords, int64_t n, int quant, uint64_t* __restrict__ keys,
    int32_t* __restrict__ vals, int64_t cap, int32_t* __restrict__ slots, int32_t* __restrict__ errflag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  CoordRow r = load_quant(coords, i, quant);
  if (!coord_in_range(r.b, r.x, r.y, r.z)) {
    atomicExch(errflag, 1);
    slots[i] = -1;
    return;
  }
  const uint64_t key = pack_key(r.b, r.x, r.y, r.z);
  uint64_t slot = hash_key(key) & (uint64_t)(cap - 1);
  for (;;) {
    unsigned long long prev = atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)kEmptyKey,
                                        (unsigned long long)key);
    if (prev == kEmptyKey || prev == key) break;
    slot = (slot + 1) & (uint64_t)(cap - 1);
  }
  atomicMin(&vals[slot], (int32_t)i);  // first occurrence wins
  slots[i] = (int32_t)slot;
}

struct FirstOccFlag {
  const int32_t* slots;
  const int32_t* vals;
  __device__ int operator()(int64_t i) const {
    int s = slots[i];
    return (s >= 0 && vals[s] == (int32_t)i) ? 1 : 0;
  }
};

struct UniqueEmit {
  const int32_t* coords;
  int quant;
  int64_t* unique_idx;
  int32_t* out_coords;
  int32_t* rank_of_row;
  __device__ void operator()(int64_t i, int64_t rank, int flag) const {
    if (!flag) return;
    unique_idx[rank] = i;
    rank_of_row[i] = (int32_t)rank;
    if (out_coords) {
      CoordRow r = load_quant(coords, i, quant);
      *reinterpret_cast<int4*>(out_coords + 4 * rank) = make_int4(r.b, r.x, r.y, r.z);
    }
  }
};

__global__ __launch_bounds__(256) void coordmap_inverse_kernel(const int32_t* __restrict__ slots,
                                                              const int32_t* __restrict__ vals,
                                                              const int32_t* __restrict__ rank_of_row, int64_t n,
                                                              int64_t* __restrict__ inverse) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int s = slots[i];
  inverse[i] = (s >= 0) ? (int64_t)rank_of_row[vals[s]] : -1;
}

// Replace "first-occurrence input row" by "map row" in the table values.
__global__ __launch_bounds__(256) void coordmap_finalize_kernel(const int32_t* __restrict__ slots,
                                                               int32_t* __restrict__ vals,
                                                               const int32_t* __restrict__ rank_of_row, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int s = slots[i];
  if (s >= 0 && vals[s] == (int32_t)i) {
    // only the owning row rewrites its slot; rank <= i so no other owner can
    // mistake the new value for its own row index unless rank == its index,
    // which only happens for the owner itself (rank_of_row[i] <= i, and rows
    // j < i own different slots).
    vals[s] = rank_of_row[i];
  }
}

__global__ void coordmap_count_kernel(const int64_t* __restrict__ scan_total, const int32_t* __restrict__ errflag,
                                      int64_t* __restrict__ n_out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) n_out[0] = (*errflag) ? -1 : *scan_total;
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kernel_map_cube_kernel(const int32_t* __restrict__ coords, int64_t n, int ts,
                                                             int ksize, const uint64_t* __restrict__ keys,
                                                             const int32_t* __restrict__ vals, int64_t cap,
                                                             int32_t* __restrict__ nbr) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n) return;
  const int4 c = *reinterpret_cast<const int4*>(coords + 4 * o);
  const int r = ksize / 2;
  int k = 0;
  for (int dz = -r; dz <= r; ++dz)
    for (int dy = -r; dy <= r; ++dy)
      for (int dx = -r; dx <= r; ++dx, ++k) {
        int v;
        if (dx == 0 && dy == 0 && dz == 0) {
          v = (int)o;
        } else {
          const int x = c.y + dx * ts, y = c.z + dy * ts, z = c.w + dz * ts;
          v = coord_in_range(c.x, x, y, z) ? table_lookup(keys, vals, cap, pack_key(c.x, x, y, z)) : -1;
        }
        nbr[(int64_t)k * n + o] = v;
      }
}

__global__ __launch_bounds__(256) void kernel_map_down2_kernel(const int32_t* __restrict__ fine, int64_t n_fine, int ts,
                                                              const int64_t* __restrict__ parent,
                                                              const int32_t* __restrict__ coarse, int64_t n_coarse,
                                                              int32_t* __restrict__ nbr2, uint8_t* __restrict__ kidx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_fine) return;
  const int64_t p = parent[i];
  const int4 cf = *reinterpret_cast<const int4*>(fine + 4 * i);
  const int4 cc = *reinterpret_cast<const int4*>(coarse + 4 * p);
  const int ox = (cf.y - cc.y) / ts, oy = (cf.z - cc.z) / ts, oz = (cf.w - cc.w) / ts;
  const int k = ox + 2 * oy + 4 * oz;
  nbr2[(int64_t)k * n_coarse + p] = (int32_t)i;
  kidx[i] = (uint8_t)k;
}

// Morton (z-order) code of the coarse cell (coord - lo) >> shift, batch index in the top bits.
__device__ inline uint32_t spread3(uint32_t v) {  // 10 bits -> every third bit
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
__global__ __launch_bounds__(256) void morton_cells_kernel(const int32_t* __restrict__ coords, int64_t n, int shift,
                                                          int lox, int loy, int loz, int bits, int64_t* __restrict__ ids) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = *reinterpret_cast<const int4*>(coords + 4 * i);
  const uint32_t x = (uint32_t)((c.y - lox) >> shift), y = (uint32_t)((c.z - loy) >> shift),
                 z = (uint32_t)((c.w - loz) >> shift);
  const uint32_t m = spread3(x) | (spread3(y) << 1) | (spread3(z) << 2);
  ids[i] = ((int64_t)c.x << (3 * bits)) | (int64_t)(m & ((1u << (3 * bits)) - 1u));
}

struct NbrFlag {
  const int32_t* nbr;
  __device__ int operator()(int64_t j) const { return nbr[j] >= 0 ? 1 : 0; }
};
struct PairEmit {
  const int32_t* nbr;
  int64_t n_out;
  int32_t* in_idx;
  int32_t* out_idx;
  int64_t* koff;
  __device__ void operator()(int64_t j, int64_t pos, int flag) const {
    const int64_t k = j / n_out, o = j - k * n_out;
    if (o == 0) koff[k] = pos;
    if (flag) {
      in_idx[pos] = nbr[j];
      out_idx[pos] = (int32_t)o;
    }
  }
};
__global__ void store_total_kernel(const int64_t* __restrict__ total, int64_t* __restrict__ dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *dst = *total;
}

}  // namespace usc

using namespace usc;

extern "C" {

const char* usc_last_error(void) { return g_err; }
int usc_abi_version(void) { return 2; }   // 2: usc_attn_bwd takes mask_bits_in_ws (round 3)

int usc_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return USC_ERR_LAUNCH;
  }
  return n;
}

int usc_voxel_floor_f64(const double* xyz, int64_t n, double voxel_size, int32_t* coords_out, usc_stream_t s) {
  USC_REQUIRE(n >= 0 && voxel_size > 0, "usc_voxel_floor_f64: bad n/voxel_size");
  if (n == 0) return USC_OK;
  USC_REQUIRE(xyz && coords_out, "usc_voxel_floor_f64: null pointer");
  hipLaunchKernelGGL(voxel_floor_kernel, dim3(stream_grid(n * 3, 256)), dim3(256), 0, as_stream(s), xyz, n * 3,
                     voxel_size, coords_out);
  USC_CHECK_LAUNCH("usc_voxel_floor_f64");
  return USC_OK;
}

int64_t usc_coordmap_capacity(int64_t n) {
  int64_t cap = 1024;
  while (cap < 2 * n) cap <<= 1;
  return cap;
}

static int64_t coordmap_fixed_ws(int64_t n) {
  // slots i32[n] | rank_of_row i32[n] | errflag (16 B)
  return align_up(n * 4, 16) * 2 + 16;
}
int64_t usc_coordmap_ws_bytes(int64_t n) { return coordmap_fixed_ws(n) + scan_ws_bytes(n); }

int usc_coordmap_build(const int32_t* coords, int64_t n, int32_t quant, uint64_t* table_keys, int32_t* table_vals,
                       int64_t cap, int64_t* unique_idx, int64_t* inverse, int32_t* out_coords, int64_t* n_out,
                       void* ws, int64_t ws_bytes, usc_stream_t s) {
  USC_REQUIRE(n >= 0, "usc_coordmap_build: negative n");
  USC_REQUIRE(table_keys && table_vals && n_out && ws, "usc_coordmap_build: null pointer");
  USC_REQUIRE(n == 0 || (coords && unique_idx && inverse), "usc_coordmap_build: null pointer");
  USC_REQUIRE(cap >= usc_coordmap_capacity(n) && (cap & (cap - 1)) == 0,
              "usc_coordmap_build: table capacity %lld too small / not a power of two for n=%lld", (long long)cap,
              (long long)n);
  USC_REQUIRE(ws_bytes >= usc_coordmap_ws_bytes(n), "usc_coordmap_build: workspace too small");
  USC_REQUIRE(n < (1ll << 31), "usc_coordmap_build: n exceeds int32 rows");
  hipStream_t st = as_stream(s);
  char* w = (char*)ws;
  int32_t* slots = (int32_t*)w;
  int32_t* rank_of_row = (int32_t*)(w + align_up(n * 4, 16));
  int32_t* errflag = (int32_t*)(w + align_up(n * 4, 16) * 2);
  void* scan_ws = w + coordmap_fixed_ws(n);
  int64_t nb = scan_num_blocks(n);

  (void)hipMemsetAsync(table_keys, 0xFF, (size_t)cap * 8, st);
  hipLaunchKernelGGL(fill_i32_kernel, dim3(stream_grid(cap, 256)), dim3(256), 0, st, table_vals, cap, 0x7fffffff);
  (void)hipMemsetAsync(errflag, 0, 16, st);
  if (n > 0) {
    const unsigned g = (unsigned)ceil_div(n, 256);
    hipLaunchKernelGGL(coordmap_insert_kernel, dim3(g), dim3(256), 0, st, coords, n, (int)quant, table_keys,
                       table_vals, cap, slots, errflag);
    FirstOccFlag f{slots, table_vals};
    UniqueEmit e{coords, (int)quant, unique_idx, out_coords, rank_of_row};
    device_exclusive_scan(f, e, n, scan_ws, st);
    hipLaunchKernelGGL(coordmap_inverse_kernel, dim3(g), dim3(256), 0, st, slots, table_vals, rank_of_row, n,
                       inverse);
    hipLaunchKernelGGL(coordmap_finalize_kernel, dim3(g), dim3(256), 0, st, slots, table_vals, rank_of_row, n);
  } else {
    (void)hipMemsetAsync(scan_ws, 0, (size_t)scan_ws_bytes(0), st);
  }
  hipLaunchKernelGGL(coordmap_count_kernel, dim3(1), dim3(64), 0, st, (const int64_t*)scan_ws + nb, errflag, n_out);
  USC_CHECK_LAUNCH("usc_coordmap_build");
  return USC_OK;
}

int usc_kernel_map_cube(const int32_t* coords, int64_t n, int32_t tensor_stride, int32_t ksize,
                        const uint64_t* table_keys, const int32_t* table_vals, int64_t cap, int32_t* nbr,
                        usc_stream_t s) {
  USC_REQUIRE(n >= 0 && tensor_stride >= 1, "usc_kernel_map_cube: bad n/tensor_stride");
  USC_REQUIRE(ksize == 3 || ksize == 1 || ksize == 5, "usc_kernel_map_cube: unsupported kernel size %d", ksize);
  if (n == 0) return USC_OK;
  USC_REQUIRE(coords && table_keys && table_vals && nbr, "usc_kernel_map_cube: null pointer");
  hipLaunchKernelGGL(kernel_map_cube_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(s), coords, n,
                     (int)tensor_stride, (int)ksize, table_keys, table_vals, cap, nbr);
  USC_CHECK_LAUNCH("usc_kernel_map_cube");
  return USC_OK;
}

int usc_kernel_map_down2(const int32_t* fine_coords, int64_t n_fine, int32_t tensor_stride, const int64_t* parent,
                         const int32_t* coarse_coords, int64_t n_coarse, int32_t* nbr2, uint8_t* kidx,
                         usc_stream_t s) {
  USC_REQUIRE(n_fine >= 0 && n_coarse >= 0 && tensor_stride >= 1, "usc_kernel_map_down2: bad sizes");
  if (n_fine == 0) return USC_OK;
  USC_REQUIRE(fine_coords && parent && coarse_coords && nbr2 && kidx, "usc_kernel_map_down2: null pointer");
  hipLaunchKernelGGL(kernel_map_down2_kernel, dim3((unsigned)ceil_div(n_fine, 256)), dim3(256), 0, as_stream(s),
                     fine_coords, n_fine, (int)tensor_stride, parent, coarse_coords, n_coarse, nbr2, kidx);
  USC_CHECK_LAUNCH("usc_kernel_map_down2");
  return USC_OK;
}

int usc_morton_cell_ids(const int32_t* coords, int64_t n, int32_t shift, int32_t lo_x, int32_t lo_y, int32_t lo_z,
                        int32_t bits_per_axis, int64_t* ids, usc_stream_t s) {
  USC_REQUIRE(n >= 0 && shift >= 0 && bits_per_axis >= 1 && bits_per_axis <= 10, "usc_morton_cell_ids: bad argument");
  if (n == 0) return USC_OK;
  USC_REQUIRE(coords && ids, "usc_morton_cell_ids: null pointer");
  hipLaunchKernelGGL(morton_cells_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, as_stream(s), coords, n,
                     (int)shift, (int)lo_x, (int)lo_y, (int)lo_z, (int)bits_per_axis, ids);
  USC_CHECK_LAUNCH("usc_morton_cell_ids");
  return USC_OK;
}

int64_t usc_rulebook_ws_bytes(int64_t K, int64_t n_out) { return scan_ws_bytes(K * n_out); }

int usc_rulebook_compact(const int32_t* nbr, int64_t K, int64_t n_out, int32_t* in_idx, int32_t* out_idx,
                         int64_t* koff, void* ws, int64_t ws_bytes, usc_stream_t s) {
  USC_REQUIRE(K >= 1 && n_out >= 0, "usc_rulebook_compact: bad K/n_out");
  USC_REQUIRE(koff && ws, "usc_rulebook_compact: null pointer");
  USC_REQUIRE(ws_bytes >= usc_rulebook_ws_bytes(K, n_out), "usc_rulebook_compact: workspace too small");
  hipStream_t st = as_stream(s);
  const int64_t total = K * n_out;
  if (total == 0) {
    (void)hipMemsetAsync(koff, 0, (size_t)(K + 1) * 8, st);
    return USC_OK;
  }
  USC_REQUIRE(nbr && in_idx && out_idx, "usc_rulebook_compact: null pointer");
  NbrFlag f{nbr};
  PairEmit e{nbr, n_out, in_idx, out_idx, koff};
  device_exclusive_scan(f, e, total, ws, st);
  hipLaunchKernelGGL(store_total_kernel, dim3(1), dim3(64), 0, st, (const int64_t*)ws + scan_num_blocks(total),
                     koff + K);
  USC_CHECK_LAUNCH("usc_rulebook_compact");
  return USC_OK;
}

}  // extern "C"
