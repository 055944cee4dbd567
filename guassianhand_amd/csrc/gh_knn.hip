// gh_knn.hip — exact K-nearest-neighbour indices of a 3-D point set against itself and the kNN-set mismatch
// ("interaction") mask of infer_one_shot.py:247-250:
//     _, idx_world, _ = knn_points(pointclouds, pointclouds, K=100)
//     _, idx_tpose, _ = knn_points(t_point, t_point, K=100)
//     mask = (idx_world == idx_tpose).sum(-1) < 10          # rank-wise comparison of the sorted neighbour lists
// knn_points is pytorch3d (third-party, not vendored in the reference): brute force, squared L2 distance accumulated as
// dist += diff*diff over x, y, z (fused by its compiler: fma(dz,dz, fma(dy,dy, dx*dx))), neighbours returned sorted by
// distance. Ties are ordered here by ascending point index (pytorch3d leaves them unspecified).
//
// MI355X form: uniform grid over the bounding box (points sorted by cell with the radix engine of gh_binning.hip, cell
// rows contiguous in x), one WAVE per query walking cubic shells of cells outwards. The running K best (distance bits <<
// 32 | index, one u64 compare = the lexicographic order) live sorted in LDS; a 64-candidate chunk is filtered against
// the current K-th key and the survivors are merged by rank counting (no sort, no atomics). The walk stops when the
// K-th distance is inside the sphere the visited cube is guaranteed to contain. Exact, deterministic.
#include "gh_internal.h"

#define GH_KNN_KMAX 128

struct GhKnnHeader {
  uint32_t n;             // element count for the radix engine
  float x0, y0, z0;       // bounding-box minimum
  float h, inv_h;         // cell edge and its reciprocal (0 when the box is degenerate)
  uint32_t pad[2];
};

struct GhKnnLayout {
  size_t header, keys_a, keys_b, vals_a, vals_b, table, cell_start, pts, total;
};

static int gh_knn_grid(int N) {                       // cells per axis: ~ a few points per occupied cell for surfaces
  int G = (int)(cbrt((double)N) * 1.2);
  return G < 2 ? 2 : (G > 128 ? 128 : G);
}

static void gh_knn_layout(int N, GhKnnLayout* L) {
  size_t off = 0;
  auto take = [&](size_t b) { size_t o = off; off += (b + 255) & ~(size_t)255; return o; };
  const size_t n = (size_t)(N > 0 ? N : 1);
  const int G = gh_knn_grid(N);
  L->header = take(sizeof(GhKnnHeader));
  L->keys_a = take(n * 4); L->keys_b = take(n * 4); L->vals_a = take(n * 4); L->vals_b = take(n * 4);
  L->table = take(gh_radix_table_words(n) * 4);
  L->cell_start = take(((size_t)G * G * G + 1) * 4);
  L->pts = take(n * 16);
  L->total = off;
}

// single block: bounding box (fixed-order min/max), cell size, element count
__global__ __launch_bounds__(1024) void gh_knn_bbox_kernel(const float* __restrict__ p, int N, int G, GhKnnHeader* __restrict__ hdr) {
  __shared__ float s_lo[3][16], s_hi[3][16];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = threadIdx.x; i < N; i += 1024) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { const float v = p[3 * i + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    for (int o = 32; o > 0; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o)); }
    if ((threadIdx.x & 63) == 0) { s_lo[a][threadIdx.x >> 6] = lo[a]; s_hi[a][threadIdx.x >> 6] = hi[a]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float ext = 0.0f;
    for (int a = 0; a < 3; ++a) {
      float l = s_lo[a][0], u = s_hi[a][0];
      for (int w = 1; w < 16; ++w) { l = fminf(l, s_lo[a][w]); u = fmaxf(u, s_hi[a][w]); }
      lo[a] = l; ext = fmaxf(ext, u - l);
    }
    hdr->n = (uint32_t)N; hdr->x0 = lo[0]; hdr->y0 = lo[1]; hdr->z0 = lo[2];
    const float h = ext / (float)G;
    hdr->h = h; hdr->inv_h = h > 0.0f ? 1.0f / h : 0.0f;
  }
}

__global__ __launch_bounds__(GH_BLOCK) void gh_knn_cell_kernel(const float* __restrict__ p, int N, int G, const GhKnnHeader* __restrict__ hdr,
                                                                uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int i = blockIdx.x * GH_BLOCK + threadIdx.x;
  if (i >= N) return;
  const float ih = hdr->inv_h;
  int c[3];
  const float o[3] = {hdr->x0, hdr->y0, hdr->z0};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float t = (p[3 * i + a] - o[a]) * ih;
    int ci = t > 0.0f ? (int)t : 0;            // NaN -> cell 0
    c[a] = ci > G - 1 ? G - 1 : ci;
  }
  keys[i] = (uint32_t)((c[2] * G + c[1]) * G + c[0]);
  vals[i] = (uint32_t)i;
}

// cell_start[c] = first sorted position whose cell id is >= c (c = 0 .. G^3)
__global__ __launch_bounds__(GH_BLOCK) void gh_knn_cell_start_kernel(const uint32_t* __restrict__ keys, int N, uint32_t ncell,
                                                                      uint32_t* __restrict__ cell_start) {
  const uint32_t c = blockIdx.x * GH_BLOCK + threadIdx.x;
  if (c > ncell) return;
  int lo = 0, hi = N;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < c) lo = mid + 1; else hi = mid; }
  cell_start[c] = (uint32_t)lo;
}

__global__ __launch_bounds__(GH_BLOCK) void gh_knn_gather_kernel(const float* __restrict__ p, const uint32_t* __restrict__ perm, int N,
                                                                  float4* __restrict__ pts) {
  const int j = blockIdx.x * GH_BLOCK + threadIdx.x;
  if (j >= N) return;
  const uint32_t i = perm[j];
  pts[j] = make_float4(p[3 * i], p[3 * i + 1], p[3 * i + 2], __uint_as_float(i));
}

__device__ __forceinline__ unsigned long long gh_readlane_u64(unsigned long long v, int l) {
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, l), hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}

// One wave per query (in cell order, so neighbouring waves share cells in L2).
__global__ __launch_bounds__(GH_BLOCK) void gh_knn_query_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ cell_of,
                                                                 const uint32_t* __restrict__ cell_start,
                                                                 const GhKnnHeader* __restrict__ hdr, int N, int K, int G,
                                                                 int32_t* __restrict__ idx_out, float* __restrict__ dist_out) {
  __shared__ unsigned long long s_list[GH_BLOCK / GH_WAVE][GH_KNN_KMAX];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = blockIdx.x * (GH_BLOCK / GH_WAVE) + w;
  if (q >= N) return;                                    // wave-uniform; the kernel has no block barriers
  unsigned long long* list = s_list[w];
  list[lane] = ~0ull; list[lane + 64] = ~0ull;
  const float4 qp = pts[q];
  const uint32_t qc = cell_of[q];
  const int cx = (int)(qc % (uint32_t)G), cy = (int)((qc / (uint32_t)G) % (uint32_t)G), cz = (int)(qc / (uint32_t)(G * G));
  const float h = hdr->h;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");

  for (int r = 0;; ++r) {
    const int side = 2 * r + 1, nslots = 2 * side * side;
    for (int sbase = 0; sbase < nslots; sbase += 64) {
      // every lane resolves one cell range of this shell: boundary rows are one run of 2r+1 cells, interior rows
      // contribute their two end cells
      const int sl = sbase + lane;
      uint32_t a = 0, b = 0;
      if (sl < nslots) {
        const int row = sl >> 1, which = sl & 1;
        const int dy = row / side - r, dz = row % side - r;
        const int y = cy + dy, z = cz + dz;
        const bool boundary = (dy == r) || (dy == -r) || (dz == r) || (dz == -r);
        int xlo, xhi;
        if (boundary) { xlo = cx - r; xhi = which ? cx - r - 1 : cx + r; }
        else { xlo = xhi = which ? cx + r : cx - r; }
        xlo = xlo < 0 ? 0 : xlo; xhi = xhi > G - 1 ? G - 1 : xhi;
        if (y >= 0 && y < G && z >= 0 && z < G && xlo <= xhi) {
          const int base = (z * G + y) * G;
          a = cell_start[base + xlo]; b = cell_start[base + xhi + 1];
        }
      }
      unsigned long long m = gh_ballot(b > a);
      while (m) {
        const int l = __builtin_ctzll(m);
        m &= m - 1;
        const int ra = (int)__builtin_amdgcn_readlane(a, l), rb = (int)__builtin_amdgcn_readlane(b, l);
        for (int jb = ra; jb < rb; jb += 64) {
          const int j = jb + lane;
          const bool valid = j < rb;
          const float4 p = pts[valid ? j : ra];
          const float dx = qp.x - p.x, dy2 = qp.y - p.y, dz2 = qp.z - p.z;
          const float d2 = fmaf(dz2, dz2, fmaf(dy2, dy2, dx * dx));
          const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | __float_as_uint(p.w);
          const unsigned long long tau = list[K - 1];
          const bool surv = valid && key < tau;
          unsigned long long sm = gh_ballot(surv);
          if (!sm) continue;
          // merge the survivors into the sorted list by rank counting
          const unsigned long long a0 = list[lane], a1 = list[lane + 64];
          int c0 = 0, c1 = 0, rank = 0;
          while (sm) {
            const int sl2 = __builtin_ctzll(sm);
            sm &= sm - 1;
            const unsigned long long sk = gh_readlane_u64(key, sl2);
            rank += sk < key; c0 += sk < a0; c1 += sk < a1;
          }
          int lb = 0;
          if (surv) {
#pragma unroll
            for (int step = 64; step >= 1; step >>= 1) if (list[lb + step - 1] < key) lb += step;
          }
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // all reads of the old list are done
          if (lane + c0 < GH_KNN_KMAX) list[lane + c0] = a0;
          if (lane + 64 + c1 < GH_KNN_KMAX) list[lane + 64 + c1] = a1;
          if (surv && rank + lb < GH_KNN_KMAX) list[rank + lb] = key;
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
      }
    }
    // every point within (r - 0.001) * h of the query lies in the cube visited so far
    const unsigned long long tau = list[K - 1];
    const float rg = r > 0 ? ((float)r - 0.001f) * h : 0.0f;
    if (tau != ~0ull && (uint32_t)(tau >> 32) <= __float_as_uint(rg * rg)) break;
    if (r >= G) break;                                             // the cube covers the whole grid
  }
  const uint32_t self = __float_as_uint(qp.w);
  for (int k = lane; k < K; k += 64) {
    const unsigned long long e = list[k];
    idx_out[(size_t)self * K + k] = (int32_t)(uint32_t)e;
    if (dist_out) dist_out[(size_t)self * K + k] = __uint_as_float((uint32_t)(e >> 32));
  }
}

// wave per point: count rank-wise equal neighbours, flag < min_same (infer_one_shot.py:249)
__global__ __launch_bounds__(GH_BLOCK) void gh_knn_mismatch_kernel(const int32_t* __restrict__ ia, const int32_t* __restrict__ ib, int N, int K,
                                                                    int min_same, uint8_t* __restrict__ mask) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * (GH_BLOCK / GH_WAVE) + w;
  if (i >= N) return;
  int same = 0;
  for (int k = lane; k < K; k += 64) same += ia[(size_t)i * K + k] == ib[(size_t)i * K + k];
  same = (int)gh_wave_sum_u32((unsigned)same);
  if (lane == 0) mask[i] = same < min_same ? 1 : 0;
}

extern "C" size_t gh_knn_workspace_bytes(int N) {
  GhKnnLayout L;
  gh_knn_layout(N, &L);
  return L.total;
}

extern "C" int gh_knn_indices(const float* points, int N, int K, int32_t* idx_out, float* dist_out, void* workspace, size_t ws_bytes,
                              void* hip_stream) {
  if (N < 0 || K < 1 || K > GH_KNN_KMAX || (N > 0 && K > N)) return GH_ERR_INVALID_ARG;
  if (N == 0) return GH_OK;
  if (!points || !idx_out || !workspace) return GH_ERR_INVALID_ARG;
  GhKnnLayout L;
  gh_knn_layout(N, &L);
  if (ws_bytes < L.total) return GH_ERR_WORKSPACE_SMALL;
  hipStream_t s = (hipStream_t)hip_stream;
  char* ws = (char*)workspace;
  const int G = gh_knn_grid(N);
  const uint32_t ncell = (uint32_t)G * G * G;
  int bits = 1;
  while ((1u << bits) < ncell) ++bits;
  (void)hipGetLastError();
  GhKnnHeader* hdr = (GhKnnHeader*)(ws + L.header);
  uint32_t* ka = (uint32_t*)(ws + L.keys_a); uint32_t* kb = (uint32_t*)(ws + L.keys_b);
  uint32_t* va = (uint32_t*)(ws + L.vals_a); uint32_t* vb = (uint32_t*)(ws + L.vals_b);
  const int nblk = (N + GH_BLOCK - 1) / GH_BLOCK;
  hipLaunchKernelGGL(gh_knn_bbox_kernel, dim3(1), dim3(1024), 0, s, points, N, G, hdr);
  hipLaunchKernelGGL(gh_knn_cell_kernel, dim3(nblk), dim3(GH_BLOCK), 0, s, points, N, G, hdr, ka, va);
  gh_radix_sort(ka, va, kb, vb, &hdr->n, (uint32_t)N, bits, (uint32_t*)(ws + L.table), s);   // result in ka / va
  hipLaunchKernelGGL(gh_knn_cell_start_kernel, dim3((ncell + 1 + GH_BLOCK - 1) / GH_BLOCK), dim3(GH_BLOCK), 0, s, ka, N, ncell,
                     (uint32_t*)(ws + L.cell_start));
  hipLaunchKernelGGL(gh_knn_gather_kernel, dim3(nblk), dim3(GH_BLOCK), 0, s, points, va, N, (float4*)(ws + L.pts));
  const int qblk = (N + GH_BLOCK / GH_WAVE - 1) / (GH_BLOCK / GH_WAVE);
  hipLaunchKernelGGL(gh_knn_query_kernel, dim3(qblk), dim3(GH_BLOCK), 0, s, (const float4*)(ws + L.pts), ka,
                     (const uint32_t*)(ws + L.cell_start), hdr, N, K, G, idx_out, dist_out);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}

extern "C" int gh_knn_mismatch_mask(const int32_t* idx_a, const int32_t* idx_b, int N, int K, int min_same, uint8_t* mask_out,
                                    void* hip_stream) {
  if (N < 0 || K < 1) return GH_ERR_INVALID_ARG;
  if (N == 0) return GH_OK;
  if (!idx_a || !idx_b || !mask_out) return GH_ERR_INVALID_ARG;
  (void)hipGetLastError();
  const int qblk = (N + GH_BLOCK / GH_WAVE - 1) / (GH_BLOCK / GH_WAVE);
  hipLaunchKernelGGL(gh_knn_mismatch_kernel, dim3(qblk), dim3(GH_BLOCK), 0, (hipStream_t)hip_stream, idx_a, idx_b, N, K, min_same, mask_out);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}
