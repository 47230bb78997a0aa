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
#include "gat_tile_common.h"

namespace bgnn {

struct TiledArgs {
  TileBlocks tb;
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

template <int HC, int C, int K, int TH, int TW>
// (8 heads: the halo's alpha_src table takes the block's LDS past a third of the CU's -- two workgroups per CU there)
// (the 16-slab instances -- 512 columns -- are not fully unrolled by hipcc: the slab's head index stays a run-time value there and the
//  coefficient array lives in scratch, 160 / 304 B per lane; rare shapes, still 3x the thread-per-node kernel they replace)
__global__ __launch_bounds__(256, (HC / C >= 8 ? 2 : 3)) void gat_aggregate_tiled_kernel(TiledArgs a) {
  static_assert(TH == TILE_H && TW == TILE_W, "one thread per cell of a 16x16 block");
  constexpr int H = HC / C;
  constexpr int HW_ = HALO_W, HR = HALO_ROWS;
  constexpr int NSLAB = HC / 32, SPH = C / 32;   // slabs, slabs per head
  using Off = StencilOffsets<K>;
  __shared__ float lds[HR * TILED_PITCH + HR + HR * H];
  float *slab = lds;
  int *hid = reinterpret_cast<int *>(lds + HR * TILED_PITCH);
  float *has = lds + HR * TILED_PITCH + HR;
  const BlockPos pos = decode_block(a.tb);
  const int tid = threadIdx.x;

  load_halo_ids<H, 256>(pos, a.node_id, a.asd, hid, has);
  __syncthreads();

  // ---- phase A: attention coefficients of this thread's cell ----------------------------------
  const int tr = tid / TW, tc = tid % TW;
  const int self_idx = (tr + 1) * HW_ + tc + 1;
  const int my = hid[self_idx];
  float alf[(K + 1) * H];
#pragma unroll
  for (int i = 0; i < (K + 1) * H; ++i) alf[i] = 0.0f;
  if (my >= 0) attention_coefficients<H, K, 0, H>(my, self_idx, hid, has, a.asd, a.eattr, a.V, a.ED, alf);

  // ---- phase B: slab by slab --------------------------------------------------------------------
  // The rows of slab s + 1 are requested (into registers) right after slab s's sums have gone to the stage, i.e. BEFORE the slab's
  // output stores: their latency runs under the BatchNorm / store pass and the two barriers around it instead of at the head of the
  // next step.  (The accumulators are dead by then, so the 11 float4 of a thread's share fit where they were.)
  constexpr int NLOAD = (HR * 8 + 255) / 256;
  float4 pre[NLOAD];
  // (every step derives its lane geometry from an OPAQUE copy of the thread id: after the full unroll hipcc otherwise shares the
  //  cell / row indices of all eight steps, keeps them alive side by side and spills them -- 16 to 76 bytes per lane at k = 8)
  auto request = [&](int s) {
    int t = tid;
    asm volatile("" : "+v"(t));
#pragma unroll
    for (int p = 0; p < NLOAD; ++p) {
      const int it = t + p * 256;
      pre[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (it < HR * 8) {
        const int row = it >> 3, q = it & 7;
        int id = hid[row];
        asm volatile("" : "+v"(id));        // re-derive the row address per slab: as loop invariants the 11 + 8 pointers
                                            // are spilled, and every reload waits for the previous slab's stores
        if (id >= 0) pre[p] = *reinterpret_cast<const float4 *>(a.xw + (int64_t)id * HC + s * 32 + q * 4);
      }
    }
  };
  request(0);
#pragma unroll
  for (int s = 0; s < NSLAB; ++s) {
    int t = tid;
    asm volatile("" : "+v"(t));
#pragma unroll
    for (int p = 0; p < NLOAD; ++p) {
      const int it = t + p * 256;
      if (it < HR * 8) *reinterpret_cast<float4 *>(slab + (it >> 3) * TILED_PITCH + (it & 7) * 4) = pre[p];
    }
    __syncthreads();
    const int hh = s / SPH;   // compile-time after the full unroll
    float4 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int b = 0; b <= K; ++b) {
      const int nidx = b == K ? self_idx : self_idx - Off::dr[b < K ? b : 0] * HW_ - Off::dc[b < K ? b : 0];
      const float alpha = alf[b * H + hh];
      const float *rp = slab + nidx * TILED_PITCH;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 x = *reinterpret_cast<const float4 *>(rp + q * 4);
        acc[q].x += alpha * x.x; acc[q].y += alpha * x.y; acc[q].z += alpha * x.z; acc[q].w += alpha * x.w;
      }
    }
    __syncthreads();                      // every gather of this slab is done: reuse it as the stage
#pragma unroll
    for (int q = 0; q < 8; ++q) *reinterpret_cast<float4 *>(slab + t * TILED_PITCH + q * 4) = acc[q];
    if (s + 1 < NSLAB) request(s + 1);
    __syncthreads();
    {
      const int q = t & 7;
      const float4 sc = *reinterpret_cast<const float4 *>(a.scale + s * 32 + q * 4);
      const float4 sh = *reinterpret_cast<const float4 *>(a.shift + s * 32 + q * 4);
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int cell = p * 32 + (t >> 3);
        int id = hid[(cell / TW + 1) * HW_ + (cell % TW) + 1];
        asm volatile("" : "+v"(id));
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
  hipLaunchKernelGGL((gat_aggregate_tiled_kernel<HC, C, K, 16, 16>), dim3(a.tb.n_blocks), dim3(256), 0, ctx->stream, a);
}

// returns BGNN_ERR_UNSUPPORTED (without setting an error) when the shape has no tiled instance
int launch_gat_aggregate_tiled(bgnn_ctx *ctx, const bgnn_graph *g, const BgnnLayer &L, int C, int ED, const float *xw,
                               const float *asd, float *out, int relu) {
  if (g->kind != 0 || (g->K != 4 && g->K != 8) || g->n_blocks2 <= 0) return BGNN_ERR_UNSUPPORTED;
  const int HC = L.heads * C;
  BGNN_TRY(ensure_edge_attrs(g));
  TiledArgs a{};
  a.tb.tiles = g->d_tiles; a.tb.items2 = g->uni_h ? nullptr : g->d_items2;
  a.tb.bh = g->bh2; a.tb.bw = g->bw2; a.tb.n_blocks = g->n_blocks2;
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
  // hidden 128 (and the widths padded to it), 8 heads of 64: the thread-per-node aggregate these shapes fell to ran at 0.8 TB/s
  BGNN_TILED_CASE(512, 128) BGNN_TILED_CASE(256, 128) BGNN_TILED_CASE(128, 128) BGNN_TILED_CASE(512, 64)
#undef BGNN_TILED_CASE
  return BGNN_ERR_UNSUPPORTED;
}

}  // namespace bgnn
