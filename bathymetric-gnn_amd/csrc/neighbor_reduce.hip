// Neighbour reductions of the non-attention backbones (reference models/gnn.py:120-143, the
// torch_geometric convolutions it instantiates with default arguments):
//   GCNConv   : out_i = sum_j d_j^-1/2 d_i^-1/2 xw_j  over in-edges j -> i plus the self loop, d = in-degree + 1
//               (add_remaining_self_loops, unit edge weights); bias / BatchNorm / ReLU folded into the epilogue
//   SAGEConv  : mean_j x_j  (scatter mean: sum / max(count, 1)); written next to a copy of x_i so that
//               lin_l(mean) + lin_r(x) is ONE GEMM over the concatenated row
//   GINConv   : sum_j x_j + (1 + eps) x_i, eps = 0
// Summation order = the order torch_geometric's scatter sees the edges: in-edges in ascending edge id (stencil
// slot order), appended self loops last.  HBM / L2 bound gathers; lanes own float4 channel groups of one node.
#include "bgnn_internal.h"

namespace bgnn {

struct ReduceArgs {
  const float *x;          // [N][D]
  const int32_t *nbr;      // ELL [N][K] (-1 = no edge) or CSR col[E]
  const int32_t *rowptr;   // CSR only
  const float *dinv;       // GCN: d^-1/2 per node
  const float *scale;      // optional epilogue: out * scale + shift
  const float *shift;
  float *out;              // row stride ldo
  float *copy_self;        // SAGE: x_i is also written here (row stride ldo), or nullptr
  const int64_t *d_m;
  int K, D, ldo, mode, relu, explicit_self_loops;
};

enum { RED_GCN = 1, RED_MEAN = 2, RED_SUM = 3 };

__global__ __launch_bounds__(256) void degree_inv_sqrt_kernel(const int32_t *nbr, const int32_t *rowptr, int K,
                                                              const int64_t *d_m, float *dinv) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= *d_m) return;
  int deg = 1;                                           // the (remaining) self loop
  if (rowptr) deg += rowptr[i + 1] - rowptr[i];
  else for (int b = 0; b < K; ++b) deg += nbr[i * K + b] >= 0 ? 1 : 0;
  dinv[i] = 1.0f / sqrtf((float)deg);                    // deg.pow(-0.5)
}

template <int LPN>                                       // lanes per node = D / 4
__global__ __launch_bounds__(256) void neighbor_reduce_kernel(ReduceArgs a) {
  constexpr int NPW = 64 / LPN;
  const int64_t M = *a.d_m;
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPN, l = lane % LPN;
  const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t i = wave_id * NPW + sub;
  if (i >= M) return;
  int64_t beg, end;
  if (a.rowptr) { beg = a.rowptr[i]; end = a.rowptr[i + 1]; }
  else { beg = i * a.K; end = beg + a.K; }
  const float4 xi = *reinterpret_cast<const float4 *>(a.x + i * a.D + l * 4);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const float di = a.mode == RED_GCN ? a.dinv[i] : 0.0f;
  int cnt = 0;
  for (int64_t p = beg; p < end; ++p) {
    const int j = a.nbr[p];
    if (j < 0) continue;
    const float4 xj = *reinterpret_cast<const float4 *>(a.x + (int64_t)j * a.D + l * 4);
    if (a.mode == RED_GCN) {
      const float w = a.dinv[j] * di;                    // deg_inv_sqrt[row] * edge_weight(=1) * deg_inv_sqrt[col]
      acc.x += w * xj.x; acc.y += w * xj.y; acc.z += w * xj.z; acc.w += w * xj.w;
    } else {
      acc.x += xj.x; acc.y += xj.y; acc.z += xj.z; acc.w += xj.w;
    }
    ++cnt;
  }
  if (a.mode == RED_GCN) {                               // the self loop, appended last
    const float w = di * di;
    acc.x += w * xi.x; acc.y += w * xi.y; acc.z += w * xi.z; acc.w += w * xi.w;
  } else {
    if (a.explicit_self_loops) {                         // GraphBuilder(include_self_loops=True): an ordinary edge i -> i, last
      acc.x += xi.x; acc.y += xi.y; acc.z += xi.z; acc.w += xi.w;
      ++cnt;
    }
    if (a.mode == RED_MEAN) {
      const float c = (float)(cnt > 0 ? cnt : 1);
      acc.x /= c; acc.y /= c; acc.z /= c; acc.w /= c;
    } else {                                             // GIN: out = propagate(...) + (1 + eps) * x_i
      acc.x += xi.x; acc.y += xi.y; acc.z += xi.z; acc.w += xi.w;
    }
  }
  if (a.scale) {
    const float4 sc = *reinterpret_cast<const float4 *>(a.scale + l * 4);
    const float4 sh = *reinterpret_cast<const float4 *>(a.shift + l * 4);
    acc.x = acc.x * sc.x + sh.x; acc.y = acc.y * sc.y + sh.y; acc.z = acc.z * sc.z + sh.z; acc.w = acc.w * sc.w + sh.w;
  }
  if (a.relu) {
    acc.x = acc.x > 0.f ? acc.x : 0.f; acc.y = acc.y > 0.f ? acc.y : 0.f;
    acc.z = acc.z > 0.f ? acc.z : 0.f; acc.w = acc.w > 0.f ? acc.w : 0.f;
  }
  *reinterpret_cast<float4 *>(a.out + i * a.ldo + l * 4) = acc;
  if (a.copy_self) *reinterpret_cast<float4 *>(a.copy_self + i * a.ldo + l * 4) = xi;
}

int launch_degree_inv_sqrt(bgnn_ctx *ctx, const bgnn_graph *g, float *dinv) {
  const int64_t rows = g->row_capacity;
  if (rows <= 0) return BGNN_OK;
  BGNN_TRY(ensure_stencil_table(g));
  ProfScope ps(ctx, BGNN_K_AGGREGATE);
  hipLaunchKernelGGL(degree_inv_sqrt_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx->stream,
                     g->d_nbr, g->kind == 0 ? nullptr : g->d_rowptr, g->K, g->d_counts, dinv);
  BGNN_HIP_CHECK(hipGetLastError());
  return BGNN_OK;
}

int launch_neighbor_reduce(bgnn_ctx *ctx, const bgnn_graph *g, int mode, const float *x, int D, const float *dinv,
                           const float *scale, const float *shift, int relu, float *out, int ldo, float *copy_self) {
  const int64_t rows = g->row_capacity;
  if (rows <= 0) return BGNN_OK;
  BGNN_REQUIRE(D == 32 || D == 64 || D == 128, "neighbor_reduce: width %d unsupported (32, 64 or 128)", D);
  BGNN_TRY(ensure_stencil_table(g));
  ReduceArgs a{x, g->d_nbr, g->kind == 0 ? nullptr : g->d_rowptr, dinv, scale, shift, out,
               copy_self, g->d_counts, g->K, D, ldo, mode, relu, g->kind == 0 ? g->include_self_loops : 0};
  ProfScope ps(ctx, BGNN_K_AGGREGATE);
  const int lpn = D / 4, npw = 64 / lpn;
  const int64_t waves = (rows + npw - 1) / npw;
  dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  if (lpn == 32) hipLaunchKernelGGL(neighbor_reduce_kernel<32>, grid, block, 0, ctx->stream, a);
  else if (lpn == 16) hipLaunchKernelGGL(neighbor_reduce_kernel<16>, grid, block, 0, ctx->stream, a);
  else hipLaunchKernelGGL(neighbor_reduce_kernel<8>, grid, block, 0, ctx->stream, a);
  BGNN_HIP_CHECK(hipGetLastError());
  return BGNN_OK;
}

}  // namespace bgnn
