// K4 (stencil form): LDS-tiled GAT gather / per-node softmax / attention-weighted aggregate.
//
// Same arithmetic as gat_aggregate.hip (torch_geometric GATConv with edge_dim, reference
// models/gnn.py:176), specialised for graphs that came from a grid: a workgroup owns a TH x TW
// block of CELLS of one tile.  The compacted rows of the block's cells and of its 1-cell halo are
// found through the node_id grid and staged through LDS one 32-channel slab at a time, so every
// neighbour row is fetched from HBM/L2 once per block (x1.27 halo re-read) instead of 9 times.
//
//   phase 0  halo node ids + alpha_src of the halo nodes -> LDS
//   phase A  thread = cell: <= K+1 logits per head, softmax -> alpha[K+1][H] in registers
//   phase B  per slab: cooperative coalesced row-slab loads -> LDS (row pitch 36 dwords: the
//            per-cell ds_read_b128 gathers are conflict-free / 2-way), thread = cell accumulates
//            9 x 8 float4, results transposed through LDS and stored as whole 128-B row segments
//            with bias + BatchNorm (folded scale/shift) + ReLU applied.
#include "bgnn_internal.h"

namespace bgnn {

template <int K> struct StencilOffsets;
template <> struct StencilOffsets<4> {   // graph_construction.py:79-81
  static constexpr int dr[4] = {-1, 1, 0, 0};
  static constexpr int dc[4] = {0, 0, -1, 1};
};
template <> struct StencilOffsets<8> {   // graph_construction.py:83-87
  static constexpr int dr[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
  static constexpr int dc[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
};

struct TiledArgs {
  const BgnnTileMeta *tiles;
  const BgnnWorkItem *items2;   // {tile, r0, c0} per block; nullptr when every tile has one shape
  int uni_h, uni_w, bh, bw;     // uniform decode: blocks per tile = bh x bw
  int n_blocks;
  const int32_t *node_id;
  const float *xw;      // [rows][HC]
  const float *asd;     // [rows][2H]
  const float *eattr;   // [rows][K][ED]
  const float *V;       // [H][ED]
  const float *scale;   // [HC]
  const float *shift;   // [HC]
  float *out;           // [rows][HC]
  int ED, relu;
};

constexpr int TILED_PITCH = 36;   // dwords per staged row slab (32 + 4 pad)

template <int HC, int C, int K, int TH, int TW>
__global__ __launch_bounds__(256, 3) void gat_aggregate_tiled_kernel(TiledArgs a) {
  static_assert(TH * TW == 256, "one thread per cell");
  constexpr int H = HC / C;
  constexpr int HW_ = TW + 2, HR = (TH + 2) * (TW + 2);
  constexpr int NSLAB = HC / 32, SPH = C / 32;   // slabs, slabs per head
  using Off = StencilOffsets<K>;
  __shared__ float lds[HR * TILED_PITCH + HR + HR * H];
  float *slab = lds;
  int *hid = reinterpret_cast<int *>(lds + HR * TILED_PITCH);
  float *has = lds + HR * TILED_PITCH + HR;

  // XCD-aware block order: consecutive work items (adjacent cell blocks, which share halo rows)
  // run on the same XCD and hit its L2
  const int nb = a.n_blocks;
  int wid;
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = nb >> 3, r = nb & 7;
    wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int tile, r0, c0;
  if (a.items2) {
    const BgnnWorkItem it = a.items2[wid];
    tile = it.tile; r0 = it.r0; c0 = it.nr;
  } else {
    const int bpt = a.bh * a.bw;
    tile = wid / bpt;
    const int rem = wid - tile * bpt;
    r0 = (rem / a.bw) * TH; c0 = (rem % a.bw) * TW;
  }
  const BgnnTileMeta t = a.tiles[tile];
  const int h = t.h, w = t.w;
  const int tid = threadIdx.x;

  // ---- phase 0: halo ids and alpha_src ------------------------------------------------------
  for (int idx = tid; idx < HR; idx += 256) {
    const int gr = r0 + idx / HW_ - 1, gc = c0 + idx % HW_ - 1;
    int id = -1;
    if (gr >= 0 && gr < h && gc >= 0 && gc < w) {
      id = a.node_id[(int64_t)t.cell_off + (int64_t)gr * w + gc];
      if (id < 0) id = -1;
    }
    hid[idx] = id;
#pragma unroll
    for (int hh = 0; hh < H; ++hh) has[idx * H + hh] = id >= 0 ? a.asd[(int64_t)id * 2 * H + hh] : 0.0f;
  }
  __syncthreads();

  // ---- phase A: attention coefficients of this thread's cell ----------------------------------
  const int tr = tid / TW, tc = tid % TW;
  const int self_idx = (tr + 1) * HW_ + tc + 1;
  const int my = hid[self_idx];
  float al[K + 1][H];
#pragma unroll
  for (int b = 0; b <= K; ++b)
#pragma unroll
    for (int hh = 0; hh < H; ++hh) al[b][hh] = 0.0f;
  if (my >= 0) {
    const int ED = a.ED;
    float ea[K][4];
    float ea_sum[4] = {0.f, 0.f, 0.f, 0.f};
    int deg = 0;
    bool present[K];
#pragma unroll
    for (int b = 0; b < K; ++b) {
      const int nidx = self_idx - Off::dr[b] * HW_ - Off::dc[b];   // slot b <- source at -offset[b]
      present[b] = hid[nidx] >= 0;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        ea[b][f] = (f < ED && present[b]) ? a.eattr[((int64_t)my * K + b) * ED + f] : 0.0f;
        if (present[b]) ea_sum[f] += ea[b][f];
      }
      deg += present[b] ? 1 : 0;
    }
    const float cnt = (float)(deg > 0 ? deg : 1);
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
      float v[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) v[f] = f < ED ? a.V[hh * ED + f] : 0.0f;
      const float ad = a.asd[(int64_t)my * 2 * H + H + hh];
      float mx = -__builtin_inff();
      float lg[K + 1];
#pragma unroll
      for (int b = 0; b < K; ++b) {
        const int nidx = self_idx - Off::dr[b] * HW_ - Off::dc[b];
        float dot = 0.0f;
#pragma unroll
        for (int f = 0; f < 4; ++f) dot += ea[b][f] * v[f];
        float x = has[nidx * H + hh] + ad + dot;
        x = x > 0.0f ? x : 0.2f * x;
        lg[b] = x;
        if (present[b]) mx = fmaxf(mx, x);
      }
      {
        float dot = 0.0f;
#pragma unroll
        for (int f = 0; f < 4; ++f) dot += (ea_sum[f] / cnt) * v[f];
        float x = has[self_idx * H + hh] + ad + dot;
        x = x > 0.0f ? x : 0.2f * x;
        lg[K] = x;
        mx = fmaxf(mx, x);
      }
      float den = 0.0f;
#pragma unroll
      for (int b = 0; b <= K; ++b) {
        const bool on = b == K ? true : present[b];
        const float p = on ? expf(lg[b] - mx) : 0.0f;
        lg[b] = p;
        den += p;
      }
      den += 1e-16f;
#pragma unroll
      for (int b = 0; b <= K; ++b) al[b][hh] = lg[b] / den;
    }
  }

  // ---- phase B: slab by slab --------------------------------------------------------------------
#pragma unroll
  for (int s = 0; s < NSLAB; ++s) {
    constexpr int NLOAD = (HR * 8 + 255) / 256;
#pragma unroll
    for (int p = 0; p < NLOAD; ++p) {
      const int it = tid + p * 256;
      if (it < HR * 8) {
        const int row = it >> 3, q = it & 7;
        const int id = hid[row];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (id >= 0) v = *reinterpret_cast<const float4 *>(a.xw + (int64_t)id * HC + s * 32 + q * 4);
        *reinterpret_cast<float4 *>(slab + row * TILED_PITCH + q * 4) = v;
      }
    }
    __syncthreads();
    const int hh = s / SPH;   // compile-time after the full unroll
    float4 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int b = 0; b <= K; ++b) {
      const int nidx = b == K ? self_idx : self_idx - Off::dr[b < K ? b : 0] * HW_ - Off::dc[b < K ? b : 0];
      const float alpha = al[b][hh];
      const float *rp = slab + nidx * TILED_PITCH;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 x = *reinterpret_cast<const float4 *>(rp + q * 4);
        acc[q].x += alpha * x.x; acc[q].y += alpha * x.y; acc[q].z += alpha * x.z; acc[q].w += alpha * x.w;
      }
    }
    __syncthreads();                      // every gather of this slab is done: reuse it as the stage
#pragma unroll
    for (int q = 0; q < 8; ++q) *reinterpret_cast<float4 *>(slab + tid * TILED_PITCH + q * 4) = acc[q];
    __syncthreads();
    {
      const int q = tid & 7;
      const float4 sc = *reinterpret_cast<const float4 *>(a.scale + s * 32 + q * 4);
      const float4 sh = *reinterpret_cast<const float4 *>(a.shift + s * 32 + q * 4);
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int cell = p * 32 + (tid >> 3);
        const int id = hid[(cell / TW + 1) * HW_ + (cell % TW) + 1];
        if (id >= 0) {
          const float4 x = *reinterpret_cast<const float4 *>(slab + cell * TILED_PITCH + q * 4);
          float4 o;
          o.x = x.x * sc.x + sh.x; o.y = x.y * sc.y + sh.y; o.z = x.z * sc.z + sh.z; o.w = x.w * sc.w + sh.w;
          if (a.relu) {
            o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f;
            o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f;
          }
          *reinterpret_cast<float4 *>(a.out + (int64_t)id * HC + s * 32 + q * 4) = o;
        }
      }
    }
    __syncthreads();                      // stage consumed before the next slab overwrites it
  }
}

template <int HC, int C, int K>
static void launch_one(bgnn_ctx *ctx, const TiledArgs &a) {
  hipLaunchKernelGGL((gat_aggregate_tiled_kernel<HC, C, K, 16, 16>), dim3(a.n_blocks), dim3(256), 0, ctx->stream, a);
}

// returns BGNN_ERR_UNSUPPORTED (without setting an error) when the shape has no tiled instance
int launch_gat_aggregate_tiled(bgnn_ctx *ctx, const bgnn_graph *g, const BgnnLayer &L, int C, int ED, const float *xw,
                               const float *asd, float *out, int relu) {
  if (g->kind != 0 || (g->K != 4 && g->K != 8) || g->n_blocks2 <= 0) return BGNN_ERR_UNSUPPORTED;
  const int HC = L.heads * C;
  TiledArgs a{};
  a.tiles = g->d_tiles; a.items2 = g->uni_h ? nullptr : g->d_items2;
  a.uni_h = g->uni_h; a.uni_w = g->uni_w; a.bh = g->bh2; a.bw = g->bw2; a.n_blocks = g->n_blocks2;
  a.node_id = g->d_node_id; a.xw = xw; a.asd = asd; a.eattr = g->d_eattr; a.V = L.V; a.scale = L.scale;
  a.shift = L.shift; a.out = out; a.ED = ED; a.relu = relu;
  ProfScope ps(ctx, BGNN_K_AGGREGATE);
#define BGNN_TILED_CASE(hc, c)                                       \
  if (HC == hc && C == c) {                                          \
    if (g->K == 8) launch_one<hc, c, 8>(ctx, a); else launch_one<hc, c, 4>(ctx, a); \
    BGNN_HIP_CHECK(hipGetLastError());                               \
    return BGNN_OK;                                                  \
  }
  BGNN_TILED_CASE(256, 64) BGNN_TILED_CASE(128, 64) BGNN_TILED_CASE(64, 64)
  BGNN_TILED_CASE(128, 32) BGNN_TILED_CASE(64, 32) BGNN_TILED_CASE(32, 32)
#undef BGNN_TILED_CASE
  return BGNN_ERR_UNSUPPORTED;
}

}  // namespace bgnn
