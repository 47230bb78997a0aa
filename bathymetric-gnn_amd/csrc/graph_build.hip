// Graph construction on the GPU (gfx950): grid tiles -> stencil graph.
//
// Replaces GraphBuilder.build_graph and helpers (reference data/graph_construction.py:91-505).
// Compiled with -ffp-contract=off: the float sequences below restate, operation by operation,
// what numpy / scipy.ndimage do on the CPU (see oracle/graph_cpu.py), so that node features
// and edge attributes come out bit-identical to the reference, not merely close.
//
// Layout in HBM (cells = all cells of all tiles, concatenated row-major; rows = compacted valid
// cells = graph nodes, in np.where order per tile, tiles in batch order):
//   node_id[cells] i32      >= 0: node id;  < 0: -(number of valid cells before it + 1)
//   cell_of_node[rows] i32
//   x8[rows][8] f32         node features, padded to 8 columns (32-byte rows)
//   local_std[rows] f32
//   nbr[rows][K] i32        slot b = node at grid offset -offset[b] (the SOURCE of the block-b
//                           edge whose target is this node), -1 if absent; ascending b is the
//                           order torch_geometric's scatter sees this node's in-edges
//   eattr[rows][K][ED] f32  attributes of that edge
#include <type_traits>
#include "bgnn_internal.h"

namespace bgnn {

struct Stencil {
  int K;
  int dr[16], dc[16];
};

static Stencil make_stencil(int connectivity) {
  Stencil s{};
  static const int o4[4][2] = {{-1, 0}, {1, 0}, {0, -1}, {0, 1}};                      // :79-81
  static const int o8[8][2] = {{-1, -1}, {-1, 0}, {-1, 1}, {0, -1}, {0, 1}, {1, -1}, {1, 0}, {1, 1}};  // :83-87
  if (connectivity == 4) {
    s.K = 4;
    for (int i = 0; i < 4; ++i) { s.dr[i] = o4[i][0]; s.dc[i] = o4[i][1]; }
  } else {
    s.K = connectivity == 16 ? 16 : 8;
    for (int i = 0; i < 8; ++i) { s.dr[i] = o8[i][0]; s.dc[i] = o8[i][1]; }
    for (int i = 8; i < s.K; ++i) { s.dr[i] = 2 * o8[i - 8][0]; s.dc[i] = 2 * o8[i - 8][1]; }
  }
  return s;
}

// ------------------------------------------------------------------------------------------
// exclusive scan of small integer values (3 kernels: reduce / spine / apply)
// ------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_PER_THREAD = 8;
constexpr int SCAN_CHUNK = SCAN_THREADS * SCAN_PER_THREAD;

struct MaskValue {
  const uint8_t *mask;
  __device__ int operator()(int64_t i) const { return mask[i] ? 1 : 0; }
};

__device__ __forceinline__ int find_tile(const BgnnTileMeta *tiles, int n_tiles, int64_t cell) {
  int lo = 0, hi = n_tiles - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if ((int64_t)tiles[mid].cell_off <= cell) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// value(i) for the edge-export scan: index space [tile][block b (K, +1 if self loops)][cell]
struct EdgeValue {
  const BgnnTileMeta *tiles;
  int n_tiles, K, KS;
  const int32_t *node_id;
  const int32_t *nbr;
  __device__ void decode(int64_t i, int &cell, int &b) const {
    int lo = 0, hi = n_tiles - 1;
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if ((int64_t)tiles[mid].cell_off * KS <= i) lo = mid; else hi = mid - 1;
    }
    const BgnnTileMeta &t = tiles[lo];
    int64_t rel = i - (int64_t)t.cell_off * KS;
    int ncell = t.h * t.w;
    b = (int)(rel / ncell);
    cell = t.cell_off + (int)(rel - (int64_t)b * ncell);
  }
  __device__ int operator()(int64_t i) const {
    int cell, b;
    decode(i, cell, b);
    int id = node_id[cell];
    if (id < 0) return 0;
    if (b >= K) return 1;
    return nbr[(int64_t)id * K + b] >= 0 ? 1 : 0;
  }
};

template <class V>
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(V val, int64_t n, int32_t *block_sums) {
  int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_PER_THREAD;
  int s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_PER_THREAD; ++j)
    if (base + j < n) s += val(base + j);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  __shared__ int wsum[SCAN_THREADS / 64];
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < SCAN_THREADS / 64; ++w) t += wsum[w];
    block_sums[blockIdx.x] = t;
  }
}

// single block: exclusive scan of block_sums in place, total -> *total (int64)
__global__ __launch_bounds__(1024) void scan_spine_kernel(int32_t *block_sums, int n_blocks, int64_t *total, int64_t *total_copy = nullptr) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n_blocks; base += 1024) {
    int i = base + threadIdx.x;
    int v = i < n_blocks ? block_sums[i] : 0;
    int incl = v;
    int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wid; ++w) woff += wsum[w];
    int carry = carry_s;
    if (i < n_blocks) block_sums[i] = carry + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *total = (int64_t)carry_s;
    if (total_copy) *total_copy = (int64_t)carry_s;
  }
}

// block-local exclusive prefix of the thread's first element; returns it
template <class V>
__device__ __forceinline__ int block_exclusive(V &val, int64_t n, int64_t base, int (&v)[SCAN_PER_THREAD]) {
  int s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_PER_THREAD; ++j) {
    v[j] = (base + j < n) ? val(base + j) : 0;
    s += v[j];
  }
  int incl = s;
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  __shared__ int wsum[SCAN_THREADS / 64];
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < wid; ++w) woff += wsum[w];
  return woff + incl - s;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_nodes_kernel(MaskValue val, int64_t n,
                                                                        const int32_t *block_off, int32_t *node_id,
                                                                        int32_t *cell_of_node) {
  int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_PER_THREAD;
  int v[SCAN_PER_THREAD];
  int p = block_off[blockIdx.x] + block_exclusive(val, n, base, v);
#pragma unroll
  for (int j = 0; j < SCAN_PER_THREAD; ++j) {
    if (base + j < n) {
      if (v[j]) {
        node_id[base + j] = p;
        cell_of_node[p] = (int32_t)(base + j);
        ++p;
      } else {
        node_id[base + j] = -(p + 1);
      }
    }
  }
}

// Small batches (<= SCAN_SMALL_BLOCKS chunks, e.g. one 256 x 256 tile): ONE launch instead of reduce + spine + apply.  Every
// block counts the valid cells of all chunks before its own by itself (<= 126 KiB of mask bytes, 16 per load) -- redundant work
// that is far cheaper than two more dependent launches on an otherwise idle chip.  The last block writes the total.
constexpr int SCAN_SMALL_BLOCKS = 64;
__device__ __forceinline__ void scan_small_nodes_body(MaskValue val, int64_t n, int32_t *node_id, int32_t *cell_of_node,
                                                      int64_t *total, int64_t *total_copy, int chunk, int n_chunks) {
  __shared__ int red[SCAN_THREADS / 64];
  __shared__ int off_s;
  const int64_t before = (int64_t)chunk * SCAN_CHUNK;                 // cells in earlier chunks (a multiple of 16)
  int c = 0;
  const uint4 *m16 = reinterpret_cast<const uint4 *>(val.mask);
  for (int64_t i = threadIdx.x; i < before / 16; i += SCAN_THREADS) {
    const uint4 q = m16[i];                                            // 16 mask bytes, each 0 or 1 (any non-zero counts)
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      c += ((w[k] & 0xffu) != 0) + ((w[k] & 0xff00u) != 0) + ((w[k] & 0xff0000u) != 0) + ((w[k] & 0xff000000u) != 0);
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < SCAN_THREADS / 64; ++w) t += red[w];
    off_s = t;
  }
  __syncthreads();
  const int64_t base = before + (int64_t)threadIdx.x * SCAN_PER_THREAD;
  int v[SCAN_PER_THREAD];
  int p = off_s + block_exclusive(val, n, base, v);
#pragma unroll
  for (int j = 0; j < SCAN_PER_THREAD; ++j) {
    if (base + j < n) {
      if (v[j]) {
        node_id[base + j] = p;
        cell_of_node[p] = (int32_t)(base + j);
        ++p;
      } else {
        node_id[base + j] = -(p + 1);
      }
    }
  }
  if (chunk == n_chunks - 1 && threadIdx.x == SCAN_THREADS - 1) {         // last thread of the last chunk
    *total = (int64_t)p;
    if (total_copy) *total_copy = (int64_t)p;
  }
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_small_nodes_kernel(MaskValue val, int64_t n, int32_t *node_id,
                                                                        int32_t *cell_of_node, int64_t *total, int64_t *total_copy) {
  scan_small_nodes_body(val, n, node_id, cell_of_node, total, total_copy, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------
// K1a: masked 5x5 box statistics, float64, same operation order as scipy.ndimage.uniform_filter
// (graph_construction.py:378-432): axis 0 then axis 1, each a zero-extended running sum
//   tmp = x[-2]+x[-1]+x[0]+x[1]+x[2];  out[0] = tmp/5;  tmp += (x[l+2] - x[l-3]); out[l] = tmp/5
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void masked_vals(const float *depth, const uint8_t *mask, int64_t i, double &v,
                                            double &c, double &q) {
  if (mask[i]) {
    double d = (double)depth[i];
    v = d; c = 1.0; q = d * d;
  } else {
    v = 0.0; c = 0.0; q = 0.0;
  }
}

// The running sums are serial along their axis by construction (bit-exactness with scipy needs that very order), so
// these kernels are latency machines: what matters is what sits ON the chain.  Here that is one float64 addition per step
// and sum: inputs are fetched one chunk ahead by independent, coalesced loads; the step is branch-free (the window's first
// output is produced by the prologue, every later step is "s += X(l+2) - X(l-3)" with zeros beyond the axis); the / 5.0 of
// uniform_filter1d and the finalisation's divisions / square root are taken off the chain (independent per element), and each
// of the three sums of a row / column has a thread of its own.  What the kernel trace of one 256 x 256 tile showed on the way
// (rocprofv3, round 4): a prefetched value must not be touched before the barrier (a select on it put the whole load latency
// back on every chunk); float64 divisions by 5.0 and by the count were most of the instructions between two barriers (now 3 + 3
// fused operations and one shared reciprocal, bit-identical); 16-wide workgroups put 16 CUs instead of 4 on a single tile.
// stats_v + stats_h of one 256 x 256 tile in the trace: 24 + 24 us before, 15 (with the scan in the same launch) + 12 us after;
// ~8 + ~5 us for a 50 000-node ragged batch.
constexpr int STATV_CH = 16;

// x / 5.0, correctly rounded, in three float64 operations instead of the ~11 of a general division (Markstein: with
// y = RN(1/5) and q = RN(x y), RN(q + (x - 5 q) y) is the correctly rounded quotient).  A non-finite x (only possible under a
// caller's own mask) gives a non-finite q, which IS the quotient (the correction would turn inf into inf - inf).
__device__ __forceinline__ double div5(double x) {
  const double q = x * 0.2;
  const double r = __builtin_fma(-5.0, q, x);
  const double c = __builtin_fma(r, 0.2, q);
  return __builtin_fabs(q) < __builtin_inf() ? c : q;
}

// Both passes are software pipelines of two roles with ONE barrier per chunk of 16 outputs:
//   workers (threads 0..255, four waves): fetch chunk i + 2 from global memory, hand chunk i + 1 to the chain through LDS, and take
//     chunk i - 1's sums back out (the horizontal pass finalises them: divisions, square root) -- stored one chunk later still,
//     so that no wait ever sees a young store (vmcnt counts loads and stores in order);
//   chain (the waves after them, thread = (sum, row or column)): run chunk i's additions, one float64 sum per thread.
// The chain wave's critical path is 16 dependent additions and their LDS traffic; everything else happens beside it.
constexpr int stats_threads(int lanes) { return 256 + (3 * lanes + 63) / 64 * 64; }

// vertical pass: COLS columns per workgroup (64; 16 when the launch would otherwise leave most CUs idle).  Outputs are produced in
// chunks of 16 rows (l = 1 + 16 i + j); the rows that ENTER those windows (l + 2) are loaded row-major (coalesced) by the workers;
// the chain parks the RAW sums (the reader divides by 5.0) in an LDS tile that the workers write out.
template <int COLS>
__device__ __forceinline__ void stats_v_body(const BgnnTileMeta *tiles, const float *depth, const uint8_t *mask, double *vs,
                                             double *vc, double *vq, int bx, int by) {
  constexpr int RPW = 256 / COLS, NK = STATV_CH / RPW;  // rows one pass of the workers covers; passes per chunk
  __shared__ float in_d[2][STATV_CH * COLS];
  __shared__ uint8_t in_m[2][STATV_CH * COLS];
  __shared__ double outt[2][3][STATV_CH * COLS];
  const BgnnTileMeta t = tiles[by];
  const int c0 = bx * COLS;
  if (c0 >= t.w) return;                               // (uniform)
  const int h = t.h, w = t.w, tid = threadIdx.x;
  const bool worker = tid < 256;
  const int lc = tid % COLS, lr = tid / COLS;          // workers: column lc, rows lr + RPW k (k < NK) of a chunk
  const int which = lr - RPW;                          // chain threads: 0 value, 1 count, 2 square (column lc)
  const bool chain = !worker && which < 3;
  const bool col_ok = c0 + lc < w;
  const int64_t base = (int64_t)t.cell_off + c0 + (col_ok ? lc : 0);
  const int nch = (h - 1 + STATV_CH - 1) / STATV_CH;   // chunk i: outputs l = 1 + 16 i + j
  float pd[NK]; uint32_t pm[NK];                       // chunk fetched ahead
  auto fetch = [&](int i) {                            // the rows entering chunk i (rows >= h read row h - 1 and are cleared)
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int r = 3 + STATV_CH * i + lr + RPW * k;
      const int64_t o = base + (int64_t)(r < h ? r : h - 1) * w;
      pd[k] = depth[o];
      pm[k] = mask[o];
    }
    asm volatile("" ::: "memory");                     // every load is issued before anything below (and none is touched before its fill)
  };
  auto fill = [&](int i) {
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int j = lr + RPW * k;
      in_d[i & 1][j * COLS + lc] = pd[k];
      in_m[i & 1][j * COLS + lc] = (3 + STATV_CH * i + j < h && pm[k]) ? (uint8_t)1 : (uint8_t)0;
    }
  };
  double o0[NK], o1[NK], o2[NK]; int64_t oo[NK];       // sums taken back out, stored by the next flush()
#pragma unroll
  for (int k = 0; k < NK; ++k) oo[k] = -1;
  auto flush = [&]() {
#pragma unroll
    for (int k = 0; k < NK; ++k)
      if (oo[k] >= 0) { vs[oo[k]] = o0[k]; vc[oo[k]] = o1[k]; vq[oo[k]] = o2[k]; oo[k] = -1; }
  };
  auto collect = [&](int i) {
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int j = lr + RPW * k, l = 1 + STATV_CH * i + j;
      if (col_ok && l < h) {
        oo[k] = base + (int64_t)l * w;
        o0[k] = outt[i & 1][0][j * COLS + lc]; o1[k] = outt[i & 1][1][j * COLS + lc]; o2[k] = outt[i & 1][2][j * COLS + lc];
      }
    }
  };
  double hx[5];                                        // chain: hx[k] = X(last entered row - k), 0 if masked
#pragma unroll
  for (int k = 0; k < 5; ++k) hx[k] = 0.0;
  double s = 0.0;
  if (worker) {
    if (nch > 0) { fetch(0); fill(0); }
    if (nch > 1) fetch(1);
  } else if (chain) {                                  // initial window: rows 0..2 in ascending order (rows < 0 contribute nothing) = output 0
#pragma unroll
    for (int r = 0; r <= 2; ++r) {
      double a = 0.0, b = 0.0, d = 0.0;
      if (r < h) masked_vals(depth, mask, base + (int64_t)r * w, a, b, d);
      const double x = which == 0 ? a : which == 1 ? b : d;
      s += x;
      hx[2 - r] = x;
    }
    if (col_ok) (which == 0 ? vs : which == 1 ? vc : vq)[base] = s;
  }
  __syncthreads();
  for (int i = 0; i <= nch; ++i) {
    if (worker) {
      if (i + 1 < nch) fill(i + 1);
      flush();                                         // chunk i - 2
      if (i + 2 < nch) fetch(i + 2);
      if (i >= 1) collect(i - 1);
    } else if (chain && i < nch) {
      float xd[STATV_CH]; uint8_t xm[STATV_CH];
#pragma unroll
      for (int j = 0; j < STATV_CH; ++j) { xm[j] = in_m[i & 1][j * COLS + lc]; xd[j] = in_d[i & 1][j * COLS + lc]; }
      double o[STATV_CH];
#pragma unroll
      for (int j = 0; j < STATV_CH; ++j) {               // row l + 2 enters (zero beyond the column), five entries back leaves
        const double d = (double)xd[j];
        const double x = xm[j] != 0 ? (which == 0 ? d : which == 1 ? 1.0 : d * d) : 0.0;
        s += (x - hx[4]);
#pragma unroll
        for (int k = 4; k > 0; --k) hx[k] = hx[k - 1];
        hx[0] = x;
        o[j] = s;
      }
#pragma unroll
      for (int j = 0; j < STATV_CH; ++j) outt[i & 1][which][j * COLS + lc] = o[j];
    }
    __syncthreads();
  }
  if (worker) flush();
}

template <int COLS>
__global__ __launch_bounds__(stats_threads(COLS)) void stats_v_kernel(const BgnnTileMeta *tiles, const float *depth,
                                                                      const uint8_t *mask, double *vs, double *vc, double *vq) {
  stats_v_body<COLS>(tiles, depth, mask, vs, vc, vq, blockIdx.x, blockIdx.y);
}

// Small batches: the compaction scan and the vertical pass read the same inputs and do not depend on each other, so they share
// ONE launch -- row 0 of the grid scans (chunk = blockIdx.x), the rows above it are the vertical pass of tile blockIdx.y - 1, and
// for a ragged batch one more row fills the canvas with -1.  Every launch taken off the latency path is its own few microseconds
// (the scan alone: 8 us for one 256 x 256 tile) plus an inter-kernel gap.
template <int COLS>
__global__ __launch_bounds__(stats_threads(COLS)) void stats_v_scan_kernel(const BgnnTileMeta *tiles, const float *depth, const uint8_t *mask,
                                                           double *vs, double *vc, double *vq, int64_t cells, int n_chunks,
                                                           int32_t *node_id, int32_t *cell_of_node, int64_t *total,
                                                           int64_t *total_copy, int4 *canvas, int64_t canvas_q) {
  static_assert(SCAN_THREADS == 256, "the scan and the canvas fill are work of the first four waves");
  if (blockIdx.y == 0 || (blockIdx.y == gridDim.y - 1 && canvas)) {
    if (threadIdx.x >= 256) return;                      // (whole waves: a finished wave does not hold a barrier up)
  }
  if (blockIdx.y == 0) {
    if ((int)blockIdx.x < n_chunks)
      scan_small_nodes_body(MaskValue{mask}, cells, node_id, cell_of_node, total, total_copy, blockIdx.x, n_chunks);
    return;
  }
  if (blockIdx.y == gridDim.y - 1 && canvas) {           // ragged batches, last row: the canvas' node ids start as -1 (16 B a store)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < canvas_q; i += (int64_t)gridDim.x * 256)
      canvas[i] = make_int4(-1, -1, -1, -1);
    return;
  }
  stats_v_body<COLS>(tiles, depth, mask, vs, vc, vq, blockIdx.x, blockIdx.y - 1);
}

// finalisation of one cell: (sums / 5.0) * 25.0 -> mean, std in float64, then float32 (graph_construction.py:408-432).  Taken by the
// horizontal pass as it writes a chunk out (it used to be a launch of its own over three float64 arrays written and read back:
// 48 B per cell and one launch -- of a dozen per ragged batch -- less)
__device__ __forceinline__ void stats_finalise(double hs, double hn, double hq2, float &out_mean, float &out_std) {
  // uniform_filter(...) * 25.0   (:408-425)
  const double sum_vals = div5(hs) * 25.0;
  const double count = div5(hn) * 25.0;
  const double sum_sq = div5(hq2) * 25.0;
  const double safe = count > 1.0 ? count : 1.0;       // np.maximum(count, 1.0)
  // two quotients by one divisor in [1, 25]: the refined reciprocal of the float64 division's own expansion (v_rcp_f64 and two
  // Newton steps) is formed once, each quotient is then q = a y corrected by (a - safe q) y -- the same operations, and the same
  // bits, as two divisions, minus their range scaling and special-case fix-up, which a divisor in this range never needs
  // (a non-finite dividend gives a non-finite q, which is the quotient)
  double y = __builtin_amdgcn_rcp(safe);
  double e = __builtin_fma(-safe, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-safe, y, 1.0);
  y = __builtin_fma(y, e, y);
  const double q1 = sum_vals * y, q2 = sum_sq * y;
  const double c1 = __builtin_fma(__builtin_fma(-safe, q1, sum_vals), y, q1);
  const double c2 = __builtin_fma(__builtin_fma(-safe, q2, sum_sq), y, q2);
  const double mean = __builtin_fabs(q1) < __builtin_inf() ? c1 : q1;
  const double mean_sq = __builtin_fabs(q2) < __builtin_inf() ? c2 : q2;
  double var = mean_sq - mean * mean;
  var = var > 0.0 ? var : 0.0;                          // np.maximum(variance, 0.0) (NaN -> NaN in numpy; inputs finite)
  out_mean = (float)mean;
  out_std = (float)sqrt(var);
}

constexpr int STAT_CH = 16;

// horizontal pass: ROWS rows per workgroup (64; 16 when the launch would otherwise leave most CUs idle).  Outputs are produced in
// chunks of 16 columns (l = 1 + 16 i + j); the inputs that ENTER those windows (columns l + 2) are loaded row-major by the workers
// (16 lanes x 8 B per row), divided by 5.0 there and handed to the chain through a triple-buffered LDS tile (chunk i + 1 being
// filled, chunk i being summed in place, chunk i - 1 being finalised), so both directions are coalesced and nothing but the
// additions is left on the serial chain.
template <int ROWS>
__device__ __forceinline__ void stats_h_body(const BgnnTileMeta *tiles, const double *vs, const double *vc, const double *vq,
                                             float *local_mean, float *local_std, int bx, int by) {
  constexpr int P = STAT_CH + 1;                       // pitch in doubles: (34 r) mod 64 banks are distinct over 32 lanes
  constexpr int NK = ROWS / 16;                        // rows per worker thread
  __shared__ double tile[3][3][ROWS * P];
  __shared__ double first[3][ROWS];
  const BgnnTileMeta t = tiles[by];
  const int r0 = bx * ROWS;
  if (r0 >= t.h) return;                               // (uniform)
  const int w = t.w, h = t.h, tid = threadIdx.x;
  const int nrow = h - r0 < ROWS ? h - r0 : ROWS;
  const int64_t base0 = (int64_t)t.cell_off + (int64_t)r0 * w;
  const bool worker = tid < 256;
  const int lr = tid >> 4, lc = tid & 15;              // workers: rows lr + 16k (k < NK), column lc of the chunk
  const int ct = tid - 256;                            // chain threads: sum ct / ROWS of row ct % ROWS
  const bool chain = !worker && ct < 3 * ROWS;
  const int which = chain ? ct / ROWS : 0, crow = chain ? ct % ROWS : 0;
  const int nch = (w - 1 + STAT_CH - 1) / STAT_CH;     // chunk i: outputs l = 1 + 16 i + j
  double pv[NK], pc[NK], pq[NK];                       // chunk fetched ahead
  auto fetch = [&](int i) {                            // the columns entering chunk i, rows r0 .. r0 + ROWS - 1 (clamped addresses)
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int rr = lr + 16 * k, cc = 3 + STAT_CH * i + lc;
      const bool ok = rr < nrow && cc < w;
      const int64_t o = base0 + (int64_t)(ok ? rr : 0) * w + (ok ? cc : 0);
      pv[k] = vs[o]; pc[k] = vc[o]; pq[k] = vq[o];      // raw: fill() clears what lies outside (any use here would wait for the load)
    }
    asm volatile("" ::: "memory");                     // every load is issued before anything below
  };
  auto fill = [&](int i) {
    double (*tl)[ROWS * P] = tile[i % 3];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int o = (lr + 16 * k) * P + lc;
      const bool ok = lr + 16 * k < nrow && 3 + STAT_CH * i + lc < w;
      tl[0][o] = ok ? div5(pv[k]) : 0.0; tl[1][o] = ok ? div5(pc[k]) : 0.0; tl[2][o] = ok ? div5(pq[k]) : 0.0;
    }
  };
  float om[NK], od[NK]; int64_t oo[NK];                // results finalised, stored by the next flush()
#pragma unroll
  for (int k = 0; k < NK; ++k) oo[k] = -1;
  auto flush = [&]() {
#pragma unroll
    for (int k = 0; k < NK; ++k)
      if (oo[k] >= 0) { local_mean[oo[k]] = om[k]; local_std[oo[k]] = od[k]; oo[k] = -1; }
  };
  auto finalise = [&](int i) {                         // row-major: 16 lanes x 4 B per row
    double (*tl)[ROWS * P] = tile[i % 3];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int rr = lr + 16 * k, cc = 1 + STAT_CH * i + lc;
      if (rr < nrow && cc < w) {
        oo[k] = base0 + (int64_t)rr * w + cc;
        stats_finalise(tl[0][rr * P + lc], tl[1][rr * P + lc], tl[2][rr * P + lc], om[k], od[k]);
      }
    }
  };
  double hx[5];                                        // chain: hx[k] = X(last entered column - k)
#pragma unroll
  for (int k = 0; k < 5; ++k) hx[k] = 0.0;
  double s = 0.0;
  if (worker) {
    if (nch > 0) { fetch(0); fill(0); }
    if (nch > 1) fetch(1);
  } else if (chain) {                                  // initial window: columns 0..2 ascending (columns < 0 contribute nothing) = output 0
    const double *src = which == 0 ? vs : which == 1 ? vc : vq;
    const int64_t base = base0 + (int64_t)(crow < nrow ? crow : 0) * w;
#pragma unroll
    for (int c = 0; c <= 2; ++c) {
      const double a = c < w ? div5(src[base + c]) : 0.0;
      s += a;
      hx[2 - c] = a;
    }
    first[which][crow] = s;
  }
  __syncthreads();
  if (tid < nrow) {                                    // output 0 of every row
    float m0, s0;
    stats_finalise(first[0][tid], first[1][tid], first[2][tid], m0, s0);
    const int64_t o = base0 + (int64_t)tid * w;
    local_mean[o] = m0; local_std[o] = s0;
  }
  for (int i = 0; i <= nch; ++i) {
    if (worker) {
      if (i + 1 < nch) fill(i + 1);
      flush();                                         // chunk i - 2
      if (i + 2 < nch) fetch(i + 2);
      if (i >= 1) finalise(i - 1);
    } else if (chain && i < nch) {
      double (*tl)[ROWS * P] = tile[i % 3];
      double x[STAT_CH];
#pragma unroll
      for (int j = 0; j < STAT_CH; ++j) x[j] = tl[which][crow * P + j];
#pragma unroll
      for (int j = 0; j < STAT_CH; ++j) {              // column l + 2 enters (zero beyond the row), five entries back leaves
        const double a1 = x[j];
        s += (a1 - hx[4]);
#pragma unroll
        for (int k = 4; k > 0; --k) hx[k] = hx[k - 1];
        hx[0] = a1;
        x[j] = s;
      }
#pragma unroll
      for (int j = 0; j < STAT_CH; ++j) tl[which][crow * P + j] = x[j];   // in place: the finaliser reads it after the barrier
    }
    __syncthreads();
  }
  if (worker) flush();
}

template <int ROWS>
__global__ __launch_bounds__(stats_threads(ROWS)) void stats_h_kernel(const BgnnTileMeta *tiles, const double *vs, const double *vc,
                                                                      const double *vq, float *local_mean, float *local_std) {
  stats_h_body<ROWS>(tiles, vs, vc, vq, local_mean, local_std, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------------
// K1b + K2: per valid cell -> node features, neighbour table, edge attributes
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float nan_to_num_f32(float v) {   // np.nan_to_num(v, nan=0.0) for float32
  if (v != v) return 0.0f;
  if (v == __builtin_inff()) return 3.4028234663852886e38f;
  if (v == -__builtin_inff()) return -3.4028234663852886e38f;
  return v;
}

struct FeatureArgs {
  const BgnnTileMeta *tiles;
  const BgnnWorkItem *items;
  const float *depth;
  const uint8_t *mask;
  const float *unc;
  const float *local_mean;
  const float *local_std;
  const int32_t *node_id;
  float *x8;
  float *node_local_std;
  int32_t *nbr;
  float *eattr;           // [rows][K][ED], or nullptr: COMPACT edge storage (below)
  // compact edge storage (the default edge feature list [distance, depth_difference, slope], K = 4 / 8 / 16): of a node's K x 3
  // attributes only the slopes are stored -- the distance of slot b depends on the tile's resolution alone (tile_dist: the lengths
  // of the x, y and diagonal unit offsets; a dilated slot is exactly twice its unit slot) and the depth difference is one float32
  // subtraction of two node depths.  The fused layer kernels rebuild the attributes from these (bit for bit); the full table is
  // expanded on demand (ensure_edge_attrs) for the export and the unfused kernels.
  float *slope;           // [rows][K]
  float *node_depth;      // [rows]
  float4 *tile_dist;      // [n_tiles]  (|dx|, |dy|, diagonal, 0) as float32
  int F;
  int feat_ids[8];
  int ED;
  int edge_ids[4];
};

// length of stencil offset (dr, dc) at resolution (rx, ry), as the float32 edge attribute (graph_construction.py:340-352)
__device__ __forceinline__ double edge_length(int dr, int dc, double rx, double ry) {
  const double dx = (double)dc * rx;                   // (tgt_c - src_c) * res_x
  const double dy = (double)dr * ry;
  return sqrt(dx * dx + dy * dy);
}
__device__ __forceinline__ float edge_length_f32(double d) { return (d != d) ? 0.0f : (float)d; }
__device__ __forceinline__ void write_tile_dist(const FeatureArgs &a, const BgnnWorkItem &it, const BgnnTileMeta &t) {
  // (every workgroup of the tile's first row band writes the same three values)
  if (a.tile_dist && threadIdx.x == 0 && it.r0 == 0)
    a.tile_dist[it.tile] = make_float4(edge_length_f32(edge_length(0, -1, t.rx, t.ry)), edge_length_f32(edge_length(-1, 0, t.rx, t.ry)),
                                       edge_length_f32(edge_length(-1, -1, t.rx, t.ry)), 0.0f);
}

__device__ __forceinline__ float filled_at(const FeatureArgs &a, int64_t i) {
  // depth_filled = nan_to_num(where(valid, depth, local_mean), nan=0)   (:281-282)
  float v = a.mask[i] ? a.depth[i] : a.local_mean[i];
  return nan_to_num_f32(v);
}

// KV: stencil width with the vectorised table stores (8 or 16, 3 edge features), 0 = generic.  A template parameter so that
// the K = 16 store path's registers (64 table values per node) do not set the occupancy of the K = 8 kernel.
template <int KV>
__global__ __launch_bounds__(256) void features_kernel(FeatureArgs a, Stencil st) {
  const BgnnWorkItem it = a.items[blockIdx.x];
  const BgnnTileMeta t = a.tiles[it.tile];
  const int h = t.h, w = t.w;
  const int ncell = it.nr * w;
  // the length of stencil offset b depends on the tile's resolution only: once per workgroup instead of a float64 sqrt per edge
  __shared__ double s_dist[16];
  if ((int)threadIdx.x < st.K && threadIdx.x < 16) s_dist[threadIdx.x] = edge_length(st.dr[threadIdx.x], st.dc[threadIdx.x], t.rx, t.ry);
  write_tile_dist(a, it, t);
  __syncthreads();
  for (int li = threadIdx.x; li < ncell; li += blockDim.x) {
    const int r = it.r0 + li / w, c = li % w;
    const int64_t tb = t.cell_off;
    const int64_t idx = tb + (int64_t)r * w + c;
    const int id = a.node_id[idx];
    if (id < 0) continue;
    // ---- node features -------------------------------------------------------------
    const float f0 = filled_at(a, idx);
    // np.gradient, unit spacing, float32 (:284): central differences, one-sided at the ends
    float gy, gx;
    if (r == 0) gy = filled_at(a, idx + w) - f0;
    else if (r == h - 1) gy = f0 - filled_at(a, idx - w);
    else gy = (filled_at(a, idx + w) - filled_at(a, idx - w)) / 2.0f;
    if (c == 0) gx = filled_at(a, idx + 1) - f0;
    else if (c == w - 1) gx = f0 - filled_at(a, idx - 1);
    else gx = (filled_at(a, idx + 1) - filled_at(a, idx - 1)) / 2.0f;
    const float gmag = sqrtf(gx * gx + gy * gy);   // contraction is off: two roundings, as numpy
    // ndimage.laplace, mode='reflect': per axis in float64, cast to float32, float32 add (:447)
    const double up = (double)(r > 0 ? filled_at(a, idx - w) : f0);
    const double dn = (double)(r < h - 1 ? filled_at(a, idx + w) : f0);
    const double lf = (double)(c > 0 ? filled_at(a, idx - 1) : f0);
    const double rt = (double)(c < w - 1 ? filled_at(a, idx + 1) : f0);
    const double c0 = (double)f0 * -2.0;
    float lap = (float)(c0 + (up + dn)) + (float)(c0 + (lf + rt));
    // zero-padded 3x3 valid count (:451-456)
    int cnt = 0;
    for (int dr = -1; dr <= 1; ++dr)
      for (int dc = -1; dc <= 1; ++dc) {
        int rr = r + dr, cc = c + dc;
        if (rr >= 0 && rr < h && cc >= 0 && cc < w) cnt += a.mask[tb + (int64_t)rr * w + cc] ? 1 : 0;
      }
    if (cnt < 3) lap = 0.0f;
    const float lstd = a.local_std[idx];
    float cand[8];
    cand[BGNN_NF_DEPTH] = a.depth[idx];
    cand[BGNN_NF_LOCAL_MEAN] = a.local_mean[idx];
    cand[BGNN_NF_LOCAL_STD] = lstd;
    cand[BGNN_NF_GRADIENT_X] = gx;
    cand[BGNN_NF_GRADIENT_Y] = gy;
    cand[BGNN_NF_GRADIENT_MAGNITUDE] = gmag;
    cand[BGNN_NF_CURVATURE] = lap;
    cand[BGNN_NF_UNCERTAINTY] = a.unc ? a.unc[idx] : 0.0f;
    float xo[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      float v = 0.0f;
      if (f < a.F) {
        const int fid = a.feat_ids[f];
#pragma unroll
        for (int k = 0; k < 8; ++k) if (fid == k) v = cand[k];
        v = nan_to_num_f32(v);
      }
      xo[f] = v;
    }
    float4 *xp = reinterpret_cast<float4 *>(a.x8 + (int64_t)id * 8);
    xp[0] = make_float4(xo[0], xo[1], xo[2], xo[3]);
    xp[1] = make_float4(xo[4], xo[5], xo[6], xo[7]);
    a.node_local_std[id] = nan_to_num_f32(lstd);
    // ---- in-edges: slot b <- source cell (r - dr[b], c - dc[b])  (:196-223, :329-376) ----
    const float dz_tgt = a.depth[idx];
    if (a.node_depth) a.node_depth[id] = dz_tgt;
    auto edge_slot = [&](int b, int &sid, float (&ev)[4], float &slope_f32) {     // slope_f32: the slot's slope whatever the list holds
      const int sr = r - st.dr[b], sc = c - st.dc[b];
      sid = -1;
      int64_t sidx = 0;
      if (sr >= 0 && sr < h && sc >= 0 && sc < w) {
        sidx = tb + (int64_t)sr * w + sc;
        sid = a.node_id[sidx];
        if (sid < 0) sid = -1;
      }
      ev[0] = ev[1] = ev[2] = ev[3] = 0.0f;
      slope_f32 = 0.0f;
      if (sid >= 0) {
        const double dist = s_dist[b];
        const float dz = dz_tgt - a.depth[sidx];          // float32 subtract
        double slope = 0.0;
        if (dist > 0.0) slope = atan((double)dz / dist) * 57.29577951308232;   // np.degrees
        slope_f32 = (slope != slope) ? 0.0f : (float)slope;
        for (int f = 0; f < a.ED; ++f) {
          const int eid = a.edge_ids[f];
          float v = 0.0f;
          if (eid == BGNN_EF_DISTANCE) { double d = dist; v = (d != d) ? 0.0f : (float)d; }
          else if (eid == BGNN_EF_DEPTH_DIFFERENCE) v = nan_to_num_f32(dz);
          else if (eid == BGNN_EF_SLOPE) v = (slope != slope) ? 0.0f : (float)slope;
          ev[f] = v;
        }
      }
    };
    // the usual shapes (3 edge features, K = 8 or 16): the node's stencil row (K ids) and attribute block (K x 3 floats) leave as
    // whole 16-byte stores instead of 4 K scattered dwords (at K = 16 the scattered form wrote 5x the bytes)
    // (eight slots at a time: at K = 16 a single pass kept 64 table values live -- 177 VGPRs, two waves per SIMD)
    auto emit_rows = [&](auto kk) {
      constexpr int KK = decltype(kk)::value;
      int4 *np = reinterpret_cast<int4 *>(a.nbr + (int64_t)id * KK);
      float4 *ep = reinterpret_cast<float4 *>(a.eattr + (int64_t)id * (3 * KK));
      float4 *sp = reinterpret_cast<float4 *>(a.slope + (int64_t)id * KK);
#pragma unroll 1
      for (int half = 0; half < KK / 8; ++half) {
        int sids[8];
        float evs[24], sls[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          float ev[4];
          edge_slot(half * 8 + b, sids[b], ev, sls[b]);
          evs[3 * b] = ev[0]; evs[3 * b + 1] = ev[1]; evs[3 * b + 2] = ev[2];
        }
        if (a.nbr) {
#pragma unroll
          for (int q = 0; q < 2; ++q) np[half * 2 + q] = make_int4(sids[4 * q], sids[4 * q + 1], sids[4 * q + 2], sids[4 * q + 3]);
        }
        if (a.eattr) {
#pragma unroll
          for (int q = 0; q < 6; ++q) ep[half * 6 + q] = make_float4(evs[4 * q], evs[4 * q + 1], evs[4 * q + 2], evs[4 * q + 3]);
        } else {                                         // compact: the slopes only
#pragma unroll
          for (int q = 0; q < 2; ++q) sp[half * 2 + q] = make_float4(sls[4 * q], sls[4 * q + 1], sls[4 * q + 2], sls[4 * q + 3]);
        }
      }
    };
    if constexpr (KV != 0) {
      emit_rows(std::integral_constant<int, KV>{});
    } else {
      for (int b = 0; b < st.K; ++b) {
        int sid;
        float ev[4], sl;
        edge_slot(b, sid, ev, sl);
        if (a.nbr) a.nbr[(int64_t)id * st.K + b] = sid;
        if (a.eattr) { for (int f = 0; f < a.ED; ++f) a.eattr[((int64_t)id * st.K + b) * a.ED + f] = ev[f]; }
        else a.slope[(int64_t)id * st.K + b] = sl;
      }
    }
  }
}

// ---- LDS-tiled form of the feature kernel for the usual shapes (K = 8 / 16, 3 edge features) -------------------------------
// The thread-per-cell form above issues ~30 (K = 8) / ~60 (K = 16) scattered global loads per cell (every stencil access goes to
// L2) and one float64 division + atan per in-edge.  This form works on chunks of 8 x 64 cells of the work item's row band:
//   1. the chunk and a halo of R cells (depth, nan_to_num'ed filled depth, node id) is staged in LDS by coalesced loads --
//      every global array is then read once per chunk (+ halo) instead of once per stencil access;
//   2. an edge and its mirror image have opposite depth differences and the same length, and atan is odd, so the slope of the
//      in-edge of slot b at cell T from source S is the negative of the slope of the in-edge of slot mirror(b) at S from T:
//      every cell computes only its FORWARD slots (source earlier in row-major order: 4 of 8 / 8 of 16) and leaves the values in
//      LDS; the other half is read back, negated, from the partner cell -- or computed in place where the partner lies outside
//      the chunk (last R rows / columns of a chunk).  4.4 instead of 8 (9.3 instead of 16) float64 atan per cell.
// Outputs are bit-identical to the thread-per-cell form: (float)(-x) == -(float)x, fl(a - b) == -fl(b - a), and the cases
// where the sign does not flip (zero or NaN depth difference -> +0) are told apart (tests/test_gpu_graph.py).
template <int KV> struct FeatStencil;
template <> struct FeatStencil<8> {
  static constexpr int dr[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
  static constexpr int dc[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
};
template <> struct FeatStencil<16> {
  static constexpr int dr[16] = {-1, -1, -1, 0, 0, 1, 1, 1, -2, -2, -2, 0, 0, 2, 2, 2};
  static constexpr int dc[16] = {-1, 0, 1, -1, 1, -1, 0, 1, -2, 0, 2, -2, 2, -2, 0, 2};
};

// COMPACT (compact edge storage, FeatureArgs): only the slopes are written -- the attribute selection by edge_ids, the float32 length
// and the nan_to_num'ed depth difference of every slot drop out of pass 2, and no id table is written
template <int KV, bool COMPACT = false>
__global__ __launch_bounds__(256, (COMPACT ? 4 : 1)) void features_tiled_kernel(FeatureArgs a, Stencil st) {
  using FS = FeatStencil<KV>;
  constexpr int R = KV == 16 ? 2 : 1, CR = 8, CW = 64, SW = CW + 2 * R, SH = CR + 2 * R, NS = SH * SW;
  constexpr int NF = KV / 2;                                    // forward slots per cell: b % 8 >= 4
  __shared__ float s_depth[NS], s_fill[NS];
  __shared__ int s_nid[NS];
  __shared__ float s_fwd[CR * CW * NF];
  __shared__ double s_dist[16];
  const BgnnWorkItem it = a.items[blockIdx.x];
  const BgnnTileMeta t = a.tiles[it.tile];
  const int h = t.h, w = t.w;
  const int64_t tb = t.cell_off;
  const int tid = threadIdx.x;
  if (tid < KV) s_dist[tid] = edge_length(FS::dr[tid], FS::dc[tid], t.rx, t.ry);   // (the compile-time table: `st` indexed by a lane would have to live in memory)
  write_tile_dist(a, it, t);
  // slope of the edge source -> target with depth difference dz = depth[target] - depth[source], exactly as the form above
  auto slope_of = [&](float dz, double dist) -> float {
    double slope = 0.0;
    if (dist > 0.0) slope = atan((double)dz / dist) * 57.29577951308232;
    return (slope != slope) ? 0.0f : (float)slope;
  };
  for (int rr0 = it.r0; rr0 < it.r0 + it.nr; rr0 += CR) {
    const int rows = min(CR, it.r0 + it.nr - rr0);
    for (int c0 = 0; c0 < w; c0 += CW) {
      const int cols = min(CW, w - c0);
      __syncthreads();                                          // the previous chunk's readers are done (first pass: s_dist visible)
      for (int i = tid; i < NS; i += 256) {
        const int gr = rr0 - R + i / SW, gc = c0 - R + i % SW;
        float d = 0.0f, f = 0.0f;
        int nid = -1;
        if (gr >= 0 && gr < h && gc >= 0 && gc < w && gr < rr0 + rows + R && gc < c0 + cols + R) {
          const int64_t gi = tb + (int64_t)gr * w + gc;
          d = a.depth[gi];
          const int id = a.node_id[gi];
          nid = id < 0 ? -1 : id;
          f = nan_to_num_f32(id >= 0 ? d : a.local_mean[gi]);    // mask[gi] <=> node_id[gi] >= 0
        }
        s_depth[i] = d; s_fill[i] = f; s_nid[i] = nid;
      }
      __syncthreads();
      // ---- pass 1: forward slopes ---------------------------------------------------------------------------------------------
#pragma unroll
      for (int k = 0; k < CR * CW / 256; ++k) {
        const int li = tid + 256 * k, lr = li / CW, lc = li % CW;
        int si = (lr + R) * SW + lc + R;
        // (opaque: si only depends on the thread, so LICM hoists the address of EVERY s_xxx[si + constant] below out of the chunk loops,
        //  one VGPR each -- dozens -- and the k = 16 instance spilled; inside the loop they fold into the DS offset fields)
        asm volatile("" : "+v"(si));
        if (lr < rows && lc < cols && s_nid[si] >= 0) {
          const float dt = s_depth[si];
#pragma unroll
          for (int j = 0; j < NF; ++j) {
            const int b = (j / 4) * 8 + 4 + (j % 4);
            const int ss = si - FS::dr[b] * SW - FS::dc[b];
            float v = 0.0f;
            if (s_nid[ss] >= 0) v = slope_of(dt - s_depth[ss], s_dist[b]);
            s_fwd[j * (CR * CW) + li] = v;
          }
        }
      }
      __syncthreads();
      // ---- pass 2: node features, stencil row, edge attributes -------------------------------------------------------------------
      // (the per-cell global operands of BOTH cells of this thread are fetched before the first cell's rows are stored: loads and
      //  stores share one in-order vmcnt queue, a load issued behind the stores would wait for HBM to take them)
      float g_lstd[CR * CW / 256], g_lmean[CR * CW / 256], g_unc[CR * CW / 256];
#pragma unroll
      for (int k = 0; k < CR * CW / 256; ++k) {
        const int li = tid + 256 * k, lr = li / CW, lc = li % CW;
        g_lstd[k] = g_lmean[k] = g_unc[k] = 0.0f;
        if (lr < rows && lc < cols && s_nid[(lr + R) * SW + lc + R] >= 0) {
          const int64_t gi = tb + (int64_t)(rr0 + lr) * w + c0 + lc;
          g_lstd[k] = a.local_std[gi]; g_lmean[k] = a.local_mean[gi];
          if (a.unc) g_unc[k] = a.unc[gi];
        }
      }
#pragma unroll
      for (int k = 0; k < CR * CW / 256; ++k) {
        const int li = tid + 256 * k, lr = li / CW, lc = li % CW;
        int si = (lr + R) * SW + lc + R;
        asm volatile("" : "+v"(si));                             // (as in pass 1)
        if (!(lr < rows && lc < cols)) continue;
        const int id = s_nid[si];
        if (id < 0) continue;
        const int r = rr0 + lr, c = c0 + lc;
        const float f0 = s_fill[si];
        float gy, gx;
        if (r == 0) gy = s_fill[si + SW] - f0;
        else if (r == h - 1) gy = f0 - s_fill[si - SW];
        else gy = (s_fill[si + SW] - s_fill[si - SW]) / 2.0f;
        if (c == 0) gx = s_fill[si + 1] - f0;
        else if (c == w - 1) gx = f0 - s_fill[si - 1];
        else gx = (s_fill[si + 1] - s_fill[si - 1]) / 2.0f;
        const float gmag = sqrtf(gx * gx + gy * gy);
        const double up = (double)(r > 0 ? s_fill[si - SW] : f0);
        const double dn = (double)(r < h - 1 ? s_fill[si + SW] : f0);
        const double lf = (double)(c > 0 ? s_fill[si - 1] : f0);
        const double rt = (double)(c < w - 1 ? s_fill[si + 1] : f0);
        const double cc0 = (double)f0 * -2.0;
        float lap = (float)(cc0 + (up + dn)) + (float)(cc0 + (lf + rt));
        int cnt = 0;
#pragma unroll
        for (int dr = -1; dr <= 1; ++dr)
#pragma unroll
          for (int dc = -1; dc <= 1; ++dc) cnt += s_nid[si + dr * SW + dc] >= 0 ? 1 : 0;    // (outside the tile: staged as -1)
        if (cnt < 3) lap = 0.0f;
        const float lstd = g_lstd[k];
        const float dz_tgt = s_depth[si];
        float cand[8];
        cand[BGNN_NF_DEPTH] = dz_tgt;
        cand[BGNN_NF_LOCAL_MEAN] = g_lmean[k];
        cand[BGNN_NF_LOCAL_STD] = lstd;
        cand[BGNN_NF_GRADIENT_X] = gx;
        cand[BGNN_NF_GRADIENT_Y] = gy;
        cand[BGNN_NF_GRADIENT_MAGNITUDE] = gmag;
        cand[BGNN_NF_CURVATURE] = lap;
        cand[BGNN_NF_UNCERTAINTY] = a.unc ? g_unc[k] : 0.0f;
        float xo[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) {
          float v = 0.0f;
          if (f < a.F) {
            const int fid = a.feat_ids[f];
#pragma unroll
            for (int q = 0; q < 8; ++q) if (fid == q) v = cand[q];
            v = nan_to_num_f32(v);
          }
          xo[f] = v;
        }
        float4 *xp = reinterpret_cast<float4 *>(a.x8 + (int64_t)id * 8);
        xp[0] = make_float4(xo[0], xo[1], xo[2], xo[3]);
        xp[1] = make_float4(xo[4], xo[5], xo[6], xo[7]);
        a.node_local_std[id] = nan_to_num_f32(lstd);
        if (a.node_depth) a.node_depth[id] = dz_tgt;
        int4 *np = reinterpret_cast<int4 *>(a.nbr + (int64_t)id * KV);
        float4 *ep = reinterpret_cast<float4 *>(a.eattr + (int64_t)id * (3 * KV));
        float4 *sp = reinterpret_cast<float4 *>(a.slope + (int64_t)id * KV);
#pragma unroll 1
        for (int half = 0; half < KV / 8; ++half) {
          int sids[8];
          float evs[24];
#pragma unroll
          for (int bb = 0; bb < 8; ++bb) {
            const int b = half * 8 + bb;
            const int sdr = half ? 2 * FS::dr[bb] : FS::dr[bb], sdc = half ? 2 * FS::dc[bb] : FS::dc[bb];
            const int ss = si - sdr * SW - sdc;
            const int sid = s_nid[ss];
            sids[bb] = sid;
            float e0 = 0.0f, e1 = 0.0f, e2 = 0.0f;
            if (sid >= 0) {
              const double dist = s_dist[b];
              const float dz = dz_tgt - s_depth[ss];
              float sl;
              if (bb >= 4) {
                sl = s_fwd[(half * 4 + (bb - 4)) * (CR * CW) + li];        // this cell's own forward slot
              } else {
                // mirrored slot: the partner is cell ss, its forward slot 7 - bb of the same half (offset negated)
                const int plr = lr - sdr, plc = lc - sdc;
                if (plr < rows && plc >= 0 && plc < cols) {                 // (plr >= lr: the partner lies below or to the right)
                  const float fv = s_fwd[(half * 4 + (3 - bb)) * (CR * CW) + plr * CW + plc];
                  sl = (!(dist > 0.0) || dz != dz || dz == 0.0f) ? 0.0f : -fv;
                } else {
                  sl = slope_of(dz, dist);
                }
              }
              if constexpr (COMPACT) {
                e2 = sl;
              } else {
                float vals[3];
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                  const int eid = a.edge_ids[f];
                  float v = 0.0f;
                  if (eid == BGNN_EF_DISTANCE) { const double d = dist; v = (d != d) ? 0.0f : (float)d; }
                  else if (eid == BGNN_EF_DEPTH_DIFFERENCE) v = nan_to_num_f32(dz);
                  else if (eid == BGNN_EF_SLOPE) v = sl;
                  vals[f] = v;
                }
                e0 = vals[0]; e1 = vals[1]; e2 = vals[2];
              }
            }
            evs[3 * bb] = e0; evs[3 * bb + 1] = e1; evs[3 * bb + 2] = e2;
          }
          if constexpr (COMPACT) {
#pragma unroll
            for (int q = 0; q < 2; ++q) sp[half * 2 + q] = make_float4(evs[12 * q + 2], evs[12 * q + 5], evs[12 * q + 8], evs[12 * q + 11]);
          } else {
            if (a.nbr) {
#pragma unroll
              for (int q = 0; q < 2; ++q) np[half * 2 + q] = make_int4(sids[4 * q], sids[4 * q + 1], sids[4 * q + 2], sids[4 * q + 3]);
            }
            if (a.eattr) {
#pragma unroll
              for (int q = 0; q < 6; ++q) ep[half * 6 + q] = make_float4(evs[4 * q], evs[4 * q + 1], evs[4 * q + 2], evs[4 * q + 3]);
            } else {
#pragma unroll
              for (int q = 0; q < 2; ++q) sp[half * 2 + q] = make_float4(evs[12 * q + 2], evs[12 * q + 5], evs[12 * q + 8], evs[12 * q + 11]);
            }
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// counts per tile (nodes, edges)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_counts_kernel(const BgnnTileMeta *tiles, const int32_t *node_id,
                                                          const int32_t *nbr, int K, int self_loops,
                                                          int64_t *tile_nodes, int64_t *tile_edges) {
  const BgnnTileMeta t = tiles[blockIdx.x];
  const int ncell = t.h * t.w;
  int nn = 0, ne = 0;
  for (int i = threadIdx.x; i < ncell; i += blockDim.x) {
    int id = node_id[t.cell_off + i];
    if (id >= 0) {
      ++nn;
      for (int b = 0; b < K; ++b) ne += nbr[(int64_t)id * K + b] >= 0 ? 1 : 0;
      ne += self_loops ? 1 : 0;
    }
  }
  for (int o = 32; o > 0; o >>= 1) { nn += __shfl_down(nn, o); ne += __shfl_down(ne, o); }
  __shared__ int sn[4], se[4];
  if ((threadIdx.x & 63) == 0) { sn[threadIdx.x >> 6] = nn; se[threadIdx.x >> 6] = ne; }
  __syncthreads();
  if (threadIdx.x == 0) {
    tile_nodes[blockIdx.x] = (int64_t)sn[0] + sn[1] + sn[2] + sn[3];
    tile_edges[blockIdx.x] = (int64_t)se[0] + se[1] + se[2] + se[3];
  }
}

// ------------------------------------------------------------------------------------------
// export to the PyG layout
// ------------------------------------------------------------------------------------------
__global__ void export_nodes_kernel(const BgnnTileMeta *tiles, int n_tiles, const int64_t *counts,
                                    const int32_t *cell_of_node, const float *x8, const float *lstd, int F,
                                    float *x, float *pos, int64_t *rows, int64_t *cols, float *local_std,
                                    int64_t *batch) {
  int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= counts[0]) return;
  int cell = cell_of_node[n];
  int ti = find_tile(tiles, n_tiles, cell);
  const BgnnTileMeta t = tiles[ti];
  int rel = cell - t.cell_off;
  int r = rel / t.w, c = rel % t.w;
  if (x) for (int f = 0; f < F; ++f) x[n * F + f] = x8[n * 8 + f];
  if (pos) { pos[n * 2] = (float)c; pos[n * 2 + 1] = (float)r; }   // pos = (col, row)  (:144-147)
  if (rows) rows[n] = r;
  if (cols) cols[n] = c;
  if (local_std) local_std[n] = lstd[n];
  if (batch) batch[n] = ti;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_edges_kernel(EdgeValue val, int64_t n,
                                                                        const int32_t *block_off,
                                                                        const int64_t *counts, const float *eattr,
                                                                        int ED, int64_t *edge_index,
                                                                        float *edge_attr) {
  int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_PER_THREAD;
  int v[SCAN_PER_THREAD];
  int64_t p = (int64_t)block_off[blockIdx.x] + block_exclusive(val, n, base, v);
  const int64_t E = counts[1];
#pragma unroll
  for (int j = 0; j < SCAN_PER_THREAD; ++j) {
    if (base + j < n && v[j]) {
      int cell, b;
      val.decode(base + j, cell, b);
      const int tgt = val.node_id[cell];
      if (b < val.K) {
        // block-b edge: src = node, tgt = node + offset[b]; stored at its target's slot b
        const int64_t slot = (int64_t)tgt * val.K + b;
        if (edge_index) { edge_index[p] = val.nbr[slot]; edge_index[E + p] = tgt; }
        if (edge_attr) for (int f = 0; f < ED; ++f) edge_attr[p * ED + f] = eattr[slot * ED + f];
      } else {  // appended self loops, attributes all 0 (:226-229, :361-364)
        if (edge_index) { edge_index[p] = tgt; edge_index[E + p] = tgt; }
        if (edge_attr) for (int f = 0; f < ED; ++f) edge_attr[p * ED + f] = 0.0f;
      }
      ++p;
    }
  }
}

__global__ void scatter_kernel(const int32_t *node_id, int64_t n_cells, const float *vals, float fill, float *grid) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_cells) return;
  int id = node_id[i];
  grid[i] = id >= 0 ? vals[id] : fill;
}

__global__ void results_to_grids_kernel(const int32_t *node_id, int64_t n_cells, const int64_t *cls,
                                        const float *conf, const float *corr, const float *lstd, float norm_floor,
                                        float *cls_grid, float *conf_grid, float *corr_grid) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_cells) return;
  int id = node_id[i];
  float a = 0.f, b = 0.f, c = 0.f;
  if (id >= 0) {
    a = (float)cls[id];
    b = conf[id];
    if (corr) {                                   // models/pipeline.py:294-307
      float s = lstd[id];
      s = s > norm_floor ? s : norm_floor;        // np.maximum(local_std_grid, FLOOR)
      c = corr[id] * s;
    }
  } else if (corr) {
    c = 0.0f * norm_floor;                        // 0.0 * max(0.0, floor) = 0.0
  }
  if (cls_grid) cls_grid[i] = a;
  if (conf_grid) conf_grid[i] = b;
  if (corr_grid) corr_grid[i] = c;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
template <class V>
static int run_scan_counts(bgnn_ctx *ctx, V val, int64_t n, int32_t **block_off_out, int64_t *total_dev, int64_t *total_copy = nullptr) {
  int n_blocks = (int)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
  if (n_blocks < 1) n_blocks = 1;
  void *bs;
  BGNN_TRY(ctx_workspace(ctx, 5, (size_t)n_blocks * sizeof(int32_t), &bs));
  hipLaunchKernelGGL(scan_reduce_kernel<V>, dim3(n_blocks), dim3(SCAN_THREADS), 0, ctx->stream, val, n,
                     (int32_t *)bs);
  hipLaunchKernelGGL(scan_spine_kernel, dim3(1), dim3(1024), 0, ctx->stream, (int32_t *)bs, n_blocks, total_dev, total_copy);
  *block_off_out = (int32_t *)bs;
  return BGNN_OK;
}

// canvas cell of every grid cell: atlas[(row0 + r) * AW + col0 + c] = node id (gutters / free space stay -1)
// (atlas_tile: the grid a canvas cell belongs to -- the fused kernels look the grid's edge lengths up by it, for cells that hold a
//  node only, so the table needs no clearing.  clear0..2: result grids the canvas walk will write valid cells of -- zero-filled here,
//  cell by cell, instead of by a fill launch of their own)
struct CanvasFill {
  const BgnnTileMeta *tiles; int n_tiles; const int32_t *pos; int atlas_w; const int32_t *node_id; int64_t cells;
  int32_t *atlas, *atlas_tile; float *clear0, *clear1, *clear2;
};
__device__ __forceinline__ void atlas_fill_body(const CanvasFill &f, int64_t block) {
  const BgnnTileMeta *tiles = f.tiles; const int n_tiles = f.n_tiles; const int32_t *pos = f.pos; const int atlas_w = f.atlas_w;
  const int32_t *node_id = f.node_id; const int64_t cells = f.cells; int32_t *atlas = f.atlas, *atlas_tile = f.atlas_tile;
  float *clear0 = f.clear0, *clear1 = f.clear1, *clear2 = f.clear2;
  const int64_t i = block * 256 + threadIdx.x;
  if (i >= cells) return;
  if (clear0) clear0[i] = 0.0f;
  if (clear1) clear1[i] = 0.0f;
  if (clear2) clear2[i] = 0.0f;
  const int id = node_id[i];
  if (id < 0) return;
  const int t = find_tile(tiles, n_tiles, i);
  const BgnnTileMeta tm = tiles[t];
  const int rel = (int)(i - tm.cell_off), r = rel / tm.w, c = rel - r * tm.w;
  const int64_t at = (int64_t)(pos[2 * t] + r) * atlas_w + pos[2 * t + 1] + c;
  atlas[at] = id;
  if (atlas_tile) atlas_tile[at] = t;
}
__global__ __launch_bounds__(256) void atlas_fill_kernel(CanvasFill f) { atlas_fill_body(f, blockIdx.x); }

// Small ragged batches: the canvas fill (needs the scan) and the statistics' horizontal pass (needs the vertical pass) both follow the
// scan + vertical-pass launch and do not depend on each other: one launch, rows [0, n_tiles) of the grid are the horizontal pass,
// the rows above them the canvas fill's blocks.
template <int ROWS>
__global__ __launch_bounds__(stats_threads(ROWS)) void stats_h_canvas_kernel(const BgnnTileMeta *tiles, const double *vs, const double *vc,
                                                                             const double *vq, float *local_mean, float *local_std,
                                                                             CanvasFill f) {
  if ((int)blockIdx.y >= f.n_tiles) {
    if (threadIdx.x >= 256) return;
    atlas_fill_body(f, (int64_t)(blockIdx.y - f.n_tiles) * gridDim.x + blockIdx.x);
    return;
  }
  stats_h_body<ROWS>(tiles, vs, vc, vq, local_mean, local_std, blockIdx.x, blockIdx.y);
}

// full edge-attribute table of a compact graph, on demand: attrs[node][b] = (length of slot b, nan_to_num(depth[node] - depth[source]),
// slope[node][b]) where the source exists, zeros elsewhere -- the values the feature kernels write in their non-compact mode
__global__ __launch_bounds__(256) void expand_edge_attrs_kernel(const BgnnTileMeta *tiles, int n_tiles, const int32_t *node_id, int64_t cells,
                                                                Stencil st, const float *slope, const float *node_depth, float *eattr,
                                                                int ED, int e0id, int e1id, int e2id, int e3id) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= cells) return;
  const int id = node_id[i];
  if (id < 0) return;
  const BgnnTileMeta t = tiles[find_tile(tiles, n_tiles, i)];
  const int li = (int)(i - t.cell_off), r = li / t.w, c = li - r * t.w;
  const float dz_tgt = node_depth[id];
  for (int b = 0; b < st.K; ++b) {
    const int sr = r - st.dr[b], sc = c - st.dc[b];
    int sid = -1;
    if (sr >= 0 && sr < t.h && sc >= 0 && sc < t.w) {
      sid = node_id[(int64_t)t.cell_off + (int64_t)sr * t.w + sc];
      if (sid < 0) sid = -1;
    }
    float e0 = 0.0f, e1 = 0.0f, e2 = 0.0f;
    if (sid >= 0) {
      e0 = edge_length_f32(edge_length(st.dr[b], st.dc[b], t.rx, t.ry));
      e1 = nan_to_num_f32(dz_tgt - node_depth[sid]);
      e2 = slope[(int64_t)id * st.K + b];
    }
    // the graph's edge feature list selects / orders the columns (graph_construction.py:355-371; unknown names are skipped upstream,
    // BGNN_EF_ZERO keeps a column of zeros)
    const int ids[4] = {e0id, e1id, e2id, e3id};
    float *e = eattr + ((int64_t)id * st.K + b) * ED;
    for (int f = 0; f < ED; ++f) e[f] = ids[f] == BGNN_EF_DISTANCE ? e0 : ids[f] == BGNN_EF_DEPTH_DIFFERENCE ? e1 : ids[f] == BGNN_EF_SLOPE ? e2 : 0.0f;
  }
}

// stencil id table on demand: nbr[node][b] = node id of the cell at -offset[b] (the SOURCE of the block-b in-edge), -1 if absent --
// exactly what the feature kernels used to write inline (graph_construction.py:196-223)
__global__ __launch_bounds__(256) void stencil_table_kernel(const BgnnTileMeta *tiles, int n_tiles, const int32_t *node_id, int64_t cells,
                                                            Stencil st, int32_t *nbr) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= cells) return;
  const int id = node_id[i];
  if (id < 0) return;
  const BgnnTileMeta t = tiles[find_tile(tiles, n_tiles, i)];
  const int li = (int)(i - t.cell_off), r = li / t.w, c = li - r * t.w;
  for (int b = 0; b < st.K; ++b) {
    const int sr = r - st.dr[b], sc = c - st.dc[b];
    int sid = -1;
    if (sr >= 0 && sr < t.h && sc >= 0 && sc < t.w) {
      sid = node_id[(int64_t)t.cell_off + (int64_t)sr * t.w + sc];
      if (sid < 0) sid = -1;
    }
    nbr[(int64_t)id * st.K + b] = sid;
  }
}

int ensure_stencil_table(const bgnn_graph *g) {
  if (g->kind != 0 || g->nbr_valid || g->total_cells <= 0) return BGNN_OK;
  bgnn_ctx *ctx = g->ctx;
  const Stencil st = make_stencil(g->K);
  hipLaunchKernelGGL(stencil_table_kernel, dim3((unsigned)((g->total_cells + 255) / 256)), dim3(256), 0, ctx->stream, g->d_tiles,
                     g->n_tiles, g->d_node_id, (int64_t)g->total_cells, st, g->d_nbr);
  BGNN_HIP_CHECK(hipGetLastError());
  g->nbr_valid = true;
  return BGNN_OK;
}

int ensure_edge_attrs(const bgnn_graph *g) {
  if (g->kind != 0 || !g->compact_edges || g->eattr_valid || g->total_cells <= 0) return BGNN_OK;
  bgnn_ctx *ctx = g->ctx;
  if (!g->d_eattr) {
    void *p = nullptr;
    BGNN_TRY(ctx->pool.alloc((size_t)g->total_cells * g->K * g->ED * sizeof(float), &p));
    g->d_eattr = (float *)p;
  }
  const Stencil st = make_stencil(g->K);
  hipLaunchKernelGGL(expand_edge_attrs_kernel, dim3((unsigned)((g->total_cells + 255) / 256)), dim3(256), 0, ctx->stream, g->d_tiles,
                     g->n_tiles, g->d_node_id, (int64_t)g->total_cells, st, g->d_slope, g->d_node_depth, g->d_eattr, g->ED,
                     g->edge_ids[0], g->edge_ids[1], g->edge_ids[2], g->edge_ids[3]);
  BGNN_HIP_CHECK(hipGetLastError());
  g->eattr_valid = true;
  return BGNN_OK;
}

int launch_graph_build(bgnn_ctx *ctx, bgnn_graph *g, const bgnn_tiles *tiles, const bgnn_graph_opts *opts) {
  const Stencil st = make_stencil(opts->connectivity);
  const int64_t cells = g->total_cells;
  int max_h = 0, max_w = 0;
  for (auto &t : g->h_tiles) { if (t.h > max_h) max_h = t.h; if (t.w > max_w) max_w = t.w; }
  // box statistics: 64 running sums per workgroup fill the chip from ~500 workgroups up; below that (a single tile: four workgroups
  // a pass) the 16-wide instances put four times the CUs to work and leave a quarter of the loads, divisions and stores between barriers
  const int64_t wide_wgs = (int64_t)((max_h + 63) / 64) * g->n_tiles;
  const bool narrow = ctx->opts.stats_narrow < 0 ? wide_wgs < 384 : ctx->opts.stats_narrow != 0;
  double *vs, *vc, *vq;
  float *lmean, *lstd;
  {
    void *p;
    BGNN_TRY(ctx_workspace(ctx, 0, (size_t)cells * sizeof(double) * 3, &p));
    vs = (double *)p; vc = vs + cells; vq = vc + cells;
    BGNN_TRY(ctx_workspace(ctx, 1, (size_t)cells * sizeof(float) * 2, &p));
    lmean = (float *)p; lstd = lmean + cells;
  }
  // 1. compaction scan (small batches: in the launch of the statistics' vertical pass)
  const int n_blocks = (int)((cells + SCAN_CHUNK - 1) / SCAN_CHUNK);
  const bool small_scan = n_blocks <= SCAN_SMALL_BLOCKS && ((uintptr_t)tiles->mask & 15) == 0;
  const bool scan_with_stats = small_scan && narrow;
  bool canvas_cleared = false;
  if (!scan_with_stats) {
    ProfScope ps(ctx, BGNN_K_SCAN);
    MaskValue mv{tiles->mask};
    if (small_scan) {
      hipLaunchKernelGGL(scan_small_nodes_kernel, dim3(n_blocks), dim3(SCAN_THREADS), 0, ctx->stream, mv, cells,
                         g->d_node_id, g->d_cell_of_node, g->d_counts, g->d_n_nodes_copy);
    } else {
      int32_t *block_off;
      BGNN_TRY(run_scan_counts(ctx, mv, cells, &block_off, g->d_counts, g->d_n_nodes_copy));
      hipLaunchKernelGGL(scan_apply_nodes_kernel, dim3(n_blocks), dim3(SCAN_THREADS), 0, ctx->stream, mv, cells,
                         block_off, g->d_node_id, g->d_cell_of_node);
    }
  } else {
    ProfScope ps(ctx, BGNN_K_STATS);
    const int gx = (max_w + 15) / 16 > n_blocks ? (max_w + 15) / 16 : n_blocks;
    const int64_t canvas_cells = g->d_atlas ? (int64_t)g->atlas_h * g->atlas_w : 0;
    canvas_cleared = g->d_atlas && canvas_cells % 4 == 0 && ((uintptr_t)g->d_atlas & 15) == 0;
    hipLaunchKernelGGL(stats_v_scan_kernel<16>, dim3(gx, g->n_tiles + 1 + (canvas_cleared ? 1 : 0)), dim3(stats_threads(16)), 0, ctx->stream,
                       g->d_tiles, tiles->depth, tiles->mask, vs, vc, vq, cells, n_blocks, g->d_node_id, g->d_cell_of_node,
                       g->d_counts, g->d_n_nodes_copy, canvas_cleared ? (int4 *)g->d_atlas : (int4 *)nullptr, canvas_cells / 4);
  }
  CanvasFill cf{g->d_tiles, g->n_tiles, g->d_atlas_pos, g->atlas_w, g->d_node_id, cells, g->d_atlas, g->d_atlas_tile_of,
                g->clear_grids[0], g->clear_grids[1], g->clear_grids[2]};
  // canvas fill in the launch of the horizontal pass (its blocks are extra grid rows: gridDim.y stays far below the 65 535 limit
  // for refinement grids; a pathological batch of very wide, flat grids keeps the fill launch of its own)
  const int64_t fill_rows64 = ((cells + 255) / 256 + (max_h + 15) / 16 - 1) / ((max_h + 15) / 16);
  const bool fill_with_stats = g->d_atlas && narrow && g->n_tiles + fill_rows64 <= 60000;
  if (g->d_atlas) {
    if (!canvas_cleared)
      BGNN_HIP_CHECK(hipMemsetAsync(g->d_atlas, 0xff, (size_t)g->atlas_h * g->atlas_w * sizeof(int32_t), ctx->stream));
    if (!fill_with_stats)
      hipLaunchKernelGGL(atlas_fill_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, ctx->stream, cf);
    g->grids_cleared = true;
  }
  // 2. box statistics
  {
    ProfScope ps(ctx, BGNN_K_STATS);
    if (narrow) {
      if (!scan_with_stats)
        hipLaunchKernelGGL(stats_v_kernel<16>, dim3((max_w + 15) / 16, g->n_tiles), dim3(stats_threads(16)), 0, ctx->stream, g->d_tiles,
                           tiles->depth, tiles->mask, vs, vc, vq);
      const int gx = (max_h + 15) / 16;
      if (fill_with_stats) {
        const int fill_rows = (int)fill_rows64;
        hipLaunchKernelGGL(stats_h_canvas_kernel<16>, dim3(gx, g->n_tiles + fill_rows), dim3(stats_threads(16)), 0, ctx->stream,
                           g->d_tiles, vs, vc, vq, lmean, lstd, cf);
      } else {
        hipLaunchKernelGGL(stats_h_kernel<16>, dim3(gx, g->n_tiles), dim3(stats_threads(16)), 0, ctx->stream, g->d_tiles,
                           vs, vc, vq, lmean, lstd);
      }
    } else {
      hipLaunchKernelGGL(stats_v_kernel<64>, dim3((max_w + 63) / 64, g->n_tiles), dim3(stats_threads(64)), 0, ctx->stream, g->d_tiles,
                         tiles->depth, tiles->mask, vs, vc, vq);
      hipLaunchKernelGGL(stats_h_kernel<64>, dim3((max_h + 63) / 64, g->n_tiles), dim3(stats_threads(64)), 0, ctx->stream, g->d_tiles,
                         vs, vc, vq, lmean, lstd);
    }
  }
  // 3. features + neighbour table + edge attributes
  {
    ProfScope ps(ctx, BGNN_K_FEATURES);
    FeatureArgs a{};
    a.tiles = g->d_tiles; a.items = g->d_items; a.depth = tiles->depth; a.mask = tiles->mask;
    a.unc = tiles->uncertainty; a.local_mean = lmean; a.local_std = lstd; a.node_id = g->d_node_id;
    a.x8 = g->d_x8; a.node_local_std = g->d_local_std; a.nbr = nullptr; a.eattr = g->d_eattr;    // (stencil id table: on demand)
    a.slope = g->d_slope; a.node_depth = g->d_node_depth; a.tile_dist = g->d_tile_dist;            // (compact: d_eattr is null here)
    g->nbr_valid = false; g->eattr_valid = !g->compact_edges;
    a.F = g->F; a.ED = g->ED;
    // final column list: requested features (uncertainty skipped when absent), then
    // uncertainty appended when given and not listed (:288-316)
    int nf = 0; bool listed_unc = false;
    for (int i = 0; i < opts->n_node_features; ++i) {
      int fid = opts->node_features[i];
      if (fid == BGNN_NF_UNCERTAINTY) { listed_unc = true; if (!tiles->uncertainty) continue; }
      a.feat_ids[nf++] = fid;
    }
    if (tiles->uncertainty && !listed_unc) a.feat_ids[nf++] = BGNN_NF_UNCERTAINTY;
    for (int i = 0; i < opts->n_edge_features; ++i) a.edge_ids[i] = opts->edge_features[i];
    // the LDS-tiled form pays where its 8 x 64 chunks are mostly full: uniform batches of tiles at least one chunk wide and tall.
    // Refinement grids of 3 .. 50 cells a side (ragged VR batches) leave it 10 % full and three barriers per chunk: measured 4x slower
    // than the thread-per-cell form there.  (features_tiled = 0: always the thread-per-cell form -- the statement the tiled one is
    // tested against; 2: the tiled form for every shape, for those tests.)
    const bool tiled = ctx->opts.features_tiled == 2 || (ctx->opts.features_tiled == 1 && g->uni_w >= 64 && g->uni_h >= 8);
    const bool compact = g->compact_edges && !a.nbr;
    // (compact: only the slopes are written, whatever the edge feature list holds)
    if (st.K == 8 && tiled && compact) hipLaunchKernelGGL((features_tiled_kernel<8, true>), dim3(g->n_items), dim3(256), 0, ctx->stream, a, st);
    else if (st.K == 16 && tiled && compact) hipLaunchKernelGGL((features_tiled_kernel<16, true>), dim3(g->n_items), dim3(256), 0, ctx->stream, a, st);
    else if (st.K == 8 && a.ED == 3 && tiled) hipLaunchKernelGGL(features_tiled_kernel<8>, dim3(g->n_items), dim3(256), 0, ctx->stream, a, st);
    else if (st.K == 16 && a.ED == 3 && tiled) hipLaunchKernelGGL(features_tiled_kernel<16>, dim3(g->n_items), dim3(256), 0, ctx->stream, a, st);
    else if (st.K == 8 && (a.ED == 3 || compact)) hipLaunchKernelGGL(features_kernel<8>, dim3(g->n_items), dim3(256), 0, ctx->stream, a, st);
    else if (st.K == 16 && (a.ED == 3 || compact)) hipLaunchKernelGGL(features_kernel<16>, dim3(g->n_items), dim3(256), 0, ctx->stream, a, st);
    else hipLaunchKernelGGL(features_kernel<0>, dim3(g->n_items), dim3(256), 0, ctx->stream, a, st);
  }
  BGNN_HIP_CHECK(hipGetLastError());
  return BGNN_OK;
}

int launch_graph_count_edges(bgnn_graph *g) {
  // per-tile node / edge counts -> host prefix offsets (synchronises)
  bgnn_ctx *ctx = g->ctx;
  BGNN_TRY(ensure_stencil_table(g));
  void *p;
  BGNN_TRY(ctx_workspace(ctx, 5, (size_t)g->n_tiles * sizeof(int64_t) * 2, &p));
  int64_t *tn = (int64_t *)p, *te = tn + g->n_tiles;
  hipLaunchKernelGGL(tile_counts_kernel, dim3(g->n_tiles), dim3(256), 0, ctx->stream, g->d_tiles, g->d_node_id,
                     g->d_nbr, g->K, g->include_self_loops, tn, te);
  BGNN_HIP_CHECK(hipGetLastError());
  return BGNN_OK;
}

int launch_graph_export(bgnn_graph *g, float *x, int64_t *edge_index, float *edge_attr, float *pos,
                        int64_t *valid_rows, int64_t *valid_cols, float *local_std, int64_t *batch) {
  bgnn_ctx *ctx = g->ctx;
  ProfScope ps(ctx, BGNN_K_EXPORT);
  const int64_t rows = g->total_cells;
  if (x || pos || valid_rows || valid_cols || local_std || batch) {
    int nb = (int)((rows + 255) / 256);
    hipLaunchKernelGGL(export_nodes_kernel, dim3(nb), dim3(256), 0, ctx->stream, g->d_tiles, g->n_tiles,
                       g->d_counts, g->d_cell_of_node, g->d_x8, g->d_local_std, g->F, x, pos, valid_rows,
                       valid_cols, local_std, batch);
  }
  if (edge_index || edge_attr) {
    BGNN_TRY(ensure_stencil_table(g));
    if (edge_attr) BGNN_TRY(ensure_edge_attrs(g));
    const int KS = g->K + (g->include_self_loops ? 1 : 0);
    EdgeValue ev{g->d_tiles, g->n_tiles, g->K, KS, g->d_node_id, g->d_nbr};
    const int64_t n = (int64_t)g->total_cells * KS;
    BGNN_REQUIRE(n < ((int64_t)1 << 31), "edge export of %lld stencil slots exceeds 2^31; export in smaller batches", (long long)n);
    int32_t *block_off;
    BGNN_TRY(run_scan_counts(ctx, ev, n, &block_off, g->d_counts + 1));
    int n_blocks = (int)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
    hipLaunchKernelGGL(scan_apply_edges_kernel, dim3(n_blocks), dim3(SCAN_THREADS), 0, ctx->stream, ev, n,
                       block_off, g->d_counts, g->d_eattr, g->ED, edge_index, edge_attr);
  }
  BGNN_HIP_CHECK(hipGetLastError());
  return BGNN_OK;
}

int launch_graph_scatter(bgnn_graph *g, const float *node_values, float fill, float *grid) {
  bgnn_ctx *ctx = g->ctx;
  ProfScope ps(ctx, BGNN_K_SCATTER);
  const int64_t n = g->total_cells;
  hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, g->d_node_id, n,
                     node_values, fill, grid);
  BGNN_HIP_CHECK(hipGetLastError());
  return BGNN_OK;
}

int launch_results_to_grids(bgnn_graph *g, const int64_t *cls, const float *conf, const float *corr,
                            float norm_floor, float *cls_grid, float *conf_grid, float *corr_grid) {
  bgnn_ctx *ctx = g->ctx;
  ProfScope ps(ctx, BGNN_K_SCATTER);
  const int64_t n = g->total_cells;
  hipLaunchKernelGGL(results_to_grids_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                     g->d_node_id, n, cls, conf, corr, g->d_local_std, norm_floor, cls_grid, conf_grid, corr_grid);
  BGNN_HIP_CHECK(hipGetLastError());
  return BGNN_OK;
}

// ------------------------------------------------------------------------------------------
// generic graphs: a PyG `Data` built elsewhere (x, edge_index, edge_attr).  CSR by target with every
// row in ascending edge-id order -- the order torch_geometric's scatter sums in -- and self loops
// dropped, as GATConv does (remove_self_loops before add_self_loops(fill_value='mean')).
// ------------------------------------------------------------------------------------------
struct DegValue {
  const int32_t *deg;
  __device__ int operator()(int64_t i) const { return deg[i]; }
};

__global__ void generic_pad_x_kernel(const float *x, int F, int64_t n, float *x8) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 8) return;
  const int64_t r = i >> 3;
  const int c = (int)(i & 7);
  x8[i] = c < F ? x[r * F + c] : 0.0f;
}

__global__ void generic_degree_kernel(const int64_t *ei, int64_t E, int64_t N, int32_t *deg) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t s = ei[e], t = ei[E + e];
  if (s != t && s >= 0 && s < N && t >= 0 && t < N) atomicAdd(deg + t, 1);
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_rowptr_kernel(DegValue val, int64_t n, const int32_t *block_off,
                                                                         int32_t *rowptr) {
  int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_PER_THREAD;
  int v[SCAN_PER_THREAD];
  int p = block_off[blockIdx.x] + block_exclusive(val, n, base, v);
#pragma unroll
  for (int j = 0; j < SCAN_PER_THREAD; ++j) {
    if (base + j < n) {
      rowptr[base + j] = p;
      p += v[j];
      if (base + j == n - 1) rowptr[n] = p;
    }
  }
}

__global__ void generic_fill_kernel(const int64_t *ei, int64_t E, int64_t N, const int32_t *rowptr, int32_t *cursor,
                                    int32_t *col, int32_t *eid) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t s = ei[e], t = ei[E + e];
  if (s != t && s >= 0 && s < N && t >= 0 && t < N) {
    const int pos = rowptr[t] + atomicAdd(cursor + t, 1);
    col[pos] = (int32_t)s;
    eid[pos] = (int32_t)e;
  }
}

// one thread per row: order the row by edge id (insertion sort; rows are short), then gather the attributes
__global__ void generic_sort_rows_kernel(int64_t N, const int32_t *rowptr, int32_t *col, int32_t *eid) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int b = rowptr[i], e = rowptr[i + 1];
  for (int p = b + 1; p < e; ++p) {
    const int ke = eid[p], kc = col[p];
    int q = p - 1;
    while (q >= b && eid[q] > ke) { eid[q + 1] = eid[q]; col[q + 1] = col[q]; --q; }
    eid[q + 1] = ke; col[q + 1] = kc;
  }
}

__global__ void generic_gather_attr_kernel(const int32_t *eid, int64_t nnz_cap, const int32_t *rowptr, int64_t N,
                                           const float *edge_attr, int ED, float *out) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nnz_cap || p >= rowptr[N]) return;
  const int64_t e = eid[p];
  for (int f = 0; f < ED; ++f) out[p * ED + f] = edge_attr[e * ED + f];
}

int launch_generic_build(bgnn_ctx *ctx, bgnn_graph *g, int64_t n_nodes, int32_t n_feat, const float *x, int64_t n_edges,
                         const int64_t *edge_index, int32_t edge_dim, const float *edge_attr) {
  DevPool &P = ctx->pool;
  const int64_t N = n_nodes, E = n_edges;
  int rc = BGNN_OK;
#define GALLOC(ptr, type, count) if (rc == BGNN_OK) { void *_p = nullptr; rc = P.alloc((size_t)((count) > 0 ? (count) : 1) * sizeof(type), &_p); ptr = (type *)_p; }
  GALLOC(g->d_counts, int64_t, 4)
  GALLOC(g->d_x8, float, N * 8)
  GALLOC(g->d_rowptr, int32_t, N + 1)
  GALLOC(g->d_nbr, int32_t, E)
  GALLOC(g->d_edge_perm, int32_t, E)
  GALLOC(g->d_eattr, float, E * edge_dim)
#undef GALLOC
  BGNN_TRY(rc);
  const int64_t counts[4] = {N, E, 0, 0};
  BGNN_TRY(ctx_upload(ctx, counts, sizeof(counts), g->d_counts));
  if (N == 0) return BGNN_OK;
  void *wsp;
  BGNN_TRY(ctx_workspace(ctx, 4, (size_t)N * 2 * sizeof(int32_t), &wsp));
  int32_t *deg = (int32_t *)wsp, *cursor = deg + N;
  BGNN_HIP_CHECK(hipMemsetAsync(deg, 0, (size_t)N * 2 * sizeof(int32_t), ctx->stream));
  BGNN_HIP_CHECK(hipMemsetAsync(g->d_rowptr, 0, (size_t)(N + 1) * sizeof(int32_t), ctx->stream));
  ProfScope ps(ctx, BGNN_K_FEATURES);
  hipLaunchKernelGGL(generic_pad_x_kernel, dim3((unsigned)((N * 8 + 255) / 256)), dim3(256), 0, ctx->stream, x, n_feat, N,
                     g->d_x8);
  if (E > 0) {
    const unsigned eb = (unsigned)((E + 255) / 256);
    hipLaunchKernelGGL(generic_degree_kernel, dim3(eb), dim3(256), 0, ctx->stream, edge_index, E, N, deg);
    DegValue dv{deg};
    int32_t *block_off;
    BGNN_TRY(run_scan_counts(ctx, dv, N, &block_off, g->d_counts + 2));
    const int n_blocks = (int)((N + SCAN_CHUNK - 1) / SCAN_CHUNK);
    hipLaunchKernelGGL(scan_apply_rowptr_kernel, dim3(n_blocks), dim3(SCAN_THREADS), 0, ctx->stream, dv, N, block_off,
                       g->d_rowptr);
    hipLaunchKernelGGL(generic_fill_kernel, dim3(eb), dim3(256), 0, ctx->stream, edge_index, E, N, g->d_rowptr, cursor,
                       g->d_nbr, g->d_edge_perm);
    hipLaunchKernelGGL(generic_sort_rows_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, N,
                       g->d_rowptr, g->d_nbr, g->d_edge_perm);
    hipLaunchKernelGGL(generic_gather_attr_kernel, dim3(eb), dim3(256), 0, ctx->stream, g->d_edge_perm, E, g->d_rowptr, N,
                       edge_attr, edge_dim, g->d_eattr);
  }
  BGNN_HIP_CHECK(hipGetLastError());
  BGNN_HIP_CHECK(hipStreamSynchronize(ctx->stream));   // the caller may free x / edge_index / edge_attr afterwards
  return BGNN_OK;
}

}  // namespace bgnn
