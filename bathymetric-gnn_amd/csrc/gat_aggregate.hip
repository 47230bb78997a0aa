// K4: GAT attention over the stencil neighbourhood -- gather, per-node softmax, weighted
// aggregate, bias + BatchNorm (folded) + ReLU -- one pass, no [E,H,C] tensor ever materialised.
//
// Restates torch_geometric GATConv.forward with edge_dim (called at reference
// models/gnn.py:176; semantics in SURVEY.md Appendix B / oracle/gat_cpu.py):
//   e_ji   = leaky_relu(a_src[j] + a_dst[i] + ea_ji . V, 0.2)          V = lin_edge^T att_edge (folded)
//   self   : ea_ii = mean of the node's incoming edge attributes (0 if none)   [fill_value='mean']
//   alpha  = exp(e - max_i) / (sum_i exp(e - max_i) + 1e-16)
//   out_i  = sum_j alpha_ji * xw[j]   (in-edges in edge order, self loop last) ; + bias ; BN ; ReLU
//
// Mapping: LPN = HC/4 lanes own one node (each lane a float4 of channels; its head = channel/C);
// a wave holds 64/LPN nodes.  Every lane computes the <= K+1 logits of its head serially (degree
// is tiny), then streams the neighbour rows with 16-byte coalesced loads.
#include "bgnn_internal.h"

namespace bgnn {

struct AggArgs {
  const float *xw;        // [N][HC]
  const float *asd;       // [N][2H]
  const int32_t *nbr;     // ELL [N][K]  or CSR col[E]
  const float *eattr;     // [N][K][ED] or [E][ED]
  const int32_t *rowptr;  // CSR only (nullptr for ELL)
  const float *V;         // [H][ED]
  const float *scale;     // [HC]
  const float *shift;     // [HC]
  float *out;             // [N][HC]
  const int64_t *d_m;
  int K, H, C, ED, relu;
  int ld, hd0, Htot;      // row stride of xw / out, first head and head count of the layer (a launch covers at most 256 columns of it)
  DropSpec drop;          // training mode: dropout on the attention coefficients (thr 0: none)
};

constexpr int AGG_MAXDEG = 16;   // ELL widths 4 / 8 / 16; longer CSR rows take the two-pass loop

template <int LPN>
__global__ __launch_bounds__(256) void gat_aggregate_kernel(AggArgs a) {
  constexpr int NPW = 64 / LPN;
  constexpr int HC = LPN * 4;
  const int64_t M = *a.d_m;
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPN, l = lane % LPN;
  const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t i = wave_id * NPW + sub;
  if (i >= M) return;
  const int H = a.Htot, ED = a.ED;                  // (asd rows and the dropout key are the LAYER's: all its heads)
  const int hh = a.hd0 + (l * 4) / a.C;
  const int HCL = a.ld;                             // row stride
  int64_t beg, end;
  if (a.rowptr) { beg = a.rowptr[i]; end = a.rowptr[i + 1]; }
  else { beg = i * a.K; end = beg + a.K; }
  float v[4];
  for (int f = 0; f < 4; ++f) v[f] = f < ED ? a.V[hh * ED + f] : 0.0f;
  const float ad = a.asd[i * 2 * H + H + hh];

  // pass 1: logits, running max, self-loop attribute = mean of incoming attributes
  float ea_sum[4] = {0.f, 0.f, 0.f, 0.f};
  int deg = 0;
  float mx = -__builtin_inff();
  float ev[AGG_MAXDEG];
  const bool small = (end - beg) <= AGG_MAXDEG;
#pragma unroll
  for (int b = 0; b < AGG_MAXDEG; ++b) {
    ev[b] = -__builtin_inff();
    if (small && beg + b < end) {
      const int j = a.nbr[beg + b];
      if (j >= 0) {
        float lg = a.asd[(int64_t)j * 2 * H + hh] + ad;
        float dot = 0.0f;
        for (int f = 0; f < ED; ++f) {
          const float e = a.eattr[(beg + b) * ED + f];
          ea_sum[f] += e;
          dot += e * v[f];
        }
        lg += dot;
        lg = lg > 0.0f ? lg : 0.2f * lg;
        ev[b] = lg;
        mx = fmaxf(mx, lg);
        ++deg;
      }
    }
  }
  if (!small) {
    for (int64_t p = beg; p < end; ++p) {
      const int j = a.nbr[p];
      if (j < 0) continue;
      float lg = a.asd[(int64_t)j * 2 * H + hh] + ad;
      float dot = 0.0f;
      for (int f = 0; f < ED; ++f) {
        const float e = a.eattr[p * ED + f];
        ea_sum[f] += e;
        dot += e * v[f];
      }
      lg += dot;
      lg = lg > 0.0f ? lg : 0.2f * lg;
      mx = fmaxf(mx, lg);
      ++deg;
    }
  }
  float self_lg;
  {
    const float cnt = (float)(deg > 0 ? deg : 1);      // scatter(..., reduce='mean'): sum / max(count, 1)
    float dot = 0.0f;
    for (int f = 0; f < ED; ++f) dot += (ea_sum[f] / cnt) * v[f];
    self_lg = a.asd[i * 2 * H + hh] + ad + dot;
    self_lg = self_lg > 0.0f ? self_lg : 0.2f * self_lg;
    mx = fmaxf(mx, self_lg);
  }
  // pass 2: exp and denominator
  float den = 0.0f;
  if (small) {
#pragma unroll
    for (int b = 0; b < AGG_MAXDEG; ++b) {
      const float p = ev[b] == -__builtin_inff() ? 0.0f : expf(ev[b] - mx);
      ev[b] = p;
      den += p;
    }
  } else {
    for (int64_t p = beg; p < end; ++p) {
      const int j = a.nbr[p];
      if (j < 0) continue;
      float lg = a.asd[(int64_t)j * 2 * H + hh] + ad;
      float dot = 0.0f;
      for (int f = 0; f < ED; ++f) dot += a.eattr[p * ED + f] * v[f];
      lg += dot;
      lg = lg > 0.0f ? lg : 0.2f * lg;
      den += expf(lg - mx);
    }
  }
  const float pself = expf(self_lg - mx);
  den += pself;
  den += 1e-16f;
  // training mode, GATConv(dropout = p): alpha of edge j -> i and head hh is kept (x 1 / (1 - p)) or zeroed after the softmax
  auto dropped = [&](float al, int64_t j) -> float {
    if (a.drop.thr == 0) return al;
    const uint64_t idx = (((uint64_t)i << 32) | (uint64_t)(uint32_t)j) * (uint64_t)H + (uint64_t)hh;
    return bgnn_drop_hash(a.drop.seed, a.drop.stream, idx) >= a.drop.thr ? al * a.drop.scale : 0.0f;
  };
  // pass 3: weighted aggregate of neighbour rows
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const float *xwl = a.xw + l * 4;
  if (small) {
#pragma unroll
    for (int b = 0; b < AGG_MAXDEG; ++b) {
      if (beg + b < end) {
        const int j = a.nbr[beg + b];
        if (j >= 0) {
          const float al = dropped(ev[b] / den, j);
          const float4 x = *reinterpret_cast<const float4 *>(xwl + (int64_t)j * HCL);
          acc.x += al * x.x; acc.y += al * x.y; acc.z += al * x.z; acc.w += al * x.w;
        }
      }
    }
  } else {
    for (int64_t p = beg; p < end; ++p) {
      const int j = a.nbr[p];
      if (j < 0) continue;
      float lg = a.asd[(int64_t)j * 2 * H + hh] + ad;
      float dot = 0.0f;
      for (int f = 0; f < ED; ++f) dot += a.eattr[p * ED + f] * v[f];
      lg += dot;
      lg = lg > 0.0f ? lg : 0.2f * lg;
      const float al = dropped(expf(lg - mx) / den, j);
      const float4 x = *reinterpret_cast<const float4 *>(xwl + (int64_t)j * HCL);
      acc.x += al * x.x; acc.y += al * x.y; acc.z += al * x.z; acc.w += al * x.w;
    }
  }
  {
    const float al = dropped(pself / den, i);
    const float4 x = *reinterpret_cast<const float4 *>(xwl + i * HCL);
    acc.x += al * x.x; acc.y += al * x.y; acc.z += al * x.z; acc.w += al * x.w;
  }
  // epilogue: (+bias, BatchNorm eval) folded into scale/shift, ReLU
  const float4 sc = *reinterpret_cast<const float4 *>(a.scale + l * 4);
  const float4 sh = *reinterpret_cast<const float4 *>(a.shift + l * 4);
  float4 o;
  o.x = acc.x * sc.x + sh.x; o.y = acc.y * sc.y + sh.y; o.z = acc.z * sc.z + sh.z; o.w = acc.w * sc.w + sh.w;
  if (a.relu) {
    o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f; o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f;
  }
  *reinterpret_cast<float4 *>(a.out + i * HCL + l * 4) = o;
}

int launch_gat_aggregate(bgnn_ctx *ctx, const bgnn_graph *g, const BgnnLayer &L, int C, int ED, const float *xw,
                         const float *asd, float *out, int relu, const DropSpec *attention_drop) {
  const int64_t max_rows = g->row_capacity;
  if (max_rows <= 0) return BGNN_OK;
  BGNN_TRY(ensure_stencil_table(g));
  BGNN_TRY(ensure_edge_attrs(g));
  ProfScope ps(ctx, BGNN_K_AGGREGATE);
  AggArgs a{};
  a.xw = xw; a.asd = asd; a.nbr = g->d_nbr; a.eattr = g->d_eattr;
  a.rowptr = g->kind == 1 ? g->d_rowptr : nullptr;
  a.V = L.V; a.scale = L.scale; a.shift = L.shift; a.out = out; a.d_m = g->d_counts;
  a.K = g->K; a.H = L.heads; a.C = C; a.ED = ED; a.relu = relu;
  if (attention_drop) a.drop = *attention_drop;
  const int HC = L.heads * C;
  a.ld = HC; a.hd0 = 0; a.Htot = L.heads;
  if (HC > 256) {
    // a layer wider than 256 columns: one launch per 256-column block (whole heads: 256 % C == 0), every block reading the layer's
    // alpha table with its first head's offset
    BGNN_REQUIRE(HC % 256 == 0 && 256 % C == 0, "gat_aggregate: heads*hidden = %d unsupported", HC);
    const int64_t waves = max_rows;                  // 64 lanes per node
    dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    for (int b = 0; b < HC / 256; ++b) {
      AggArgs ab = a;
      ab.xw = xw + b * 256; ab.out = out + b * 256; ab.scale = L.scale + b * 256; ab.shift = L.shift + b * 256;
      ab.hd0 = b * (256 / C);
      hipLaunchKernelGGL(gat_aggregate_kernel<64>, grid, block, 0, ctx->stream, ab);
      BGNN_HIP_CHECK(hipGetLastError());
    }
    return BGNN_OK;
  }
  const int LPN = HC / 4;
  const int npw = 64 / LPN;
  const int64_t waves = (max_rows + npw - 1) / npw;
  dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  switch (LPN) {
    case 64: hipLaunchKernelGGL(gat_aggregate_kernel<64>, grid, block, 0, ctx->stream, a); break;
    case 32: hipLaunchKernelGGL(gat_aggregate_kernel<32>, grid, block, 0, ctx->stream, a); break;
    case 16: hipLaunchKernelGGL(gat_aggregate_kernel<16>, grid, block, 0, ctx->stream, a); break;
    case 8: hipLaunchKernelGGL(gat_aggregate_kernel<8>, grid, block, 0, ctx->stream, a); break;
    default:
      set_error("gat_aggregate: unsupported heads*hidden = %d", HC);
      return BGNN_ERR_UNSUPPORTED;
  }
  BGNN_HIP_CHECK(hipGetLastError());
  return BGNN_OK;
}

}  // namespace bgnn
