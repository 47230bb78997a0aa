// Shared device code of the LDS-tiled stencil kernels (gat_aggregate_tiled.hip, gat_layer_fused.hip):
// block -> (tile, 16x16 cell block) decode, halo node ids, per-cell attention coefficients.
#pragma once
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

template <> struct StencilOffsets<16> {  // "16-dilated" (BASELINE config 3's k = 16; not in the reference): the 8 base offsets, then the same x 2
  static constexpr int dr[16] = {-1, -1, -1, 0, 0, 1, 1, 1, -2, -2, -2, 0, 0, 2, 2, 2};
  static constexpr int dc[16] = {-1, 0, 1, -1, 1, -1, 0, 1, -2, 0, 2, -2, 2, -2, 0, 2};
};

constexpr int TILE_W = 16;                    // cell blocks are TH x 16 (TH = 16: tiled aggregate, TH = 8: fused layer)
constexpr int HALO_W = TILE_W + 2;            // 18
constexpr int TILED_PITCH = 36;               // dwords per staged 32-channel row slab (32 + 4 pad)
constexpr int TILE_H = 16, HALO_ROWS = (TILE_H + 2) * HALO_W;   // geometry of the 16x16 tiled aggregate

// halo-index offset (dr*HALO_W + dc) of stencil slot b, computed arithmetically for runtime b;
// b == K is the self loop (offset 0)
template <int K>
__device__ __forceinline__ int slot_halo_offset(int b) {
  if (b >= K) return 0;
  if (K == 8) {
    const int k = b < 4 ? b : b + 1;            // position in the 3x3 stencil, centre skipped
    return (k / 3 - 1) * (TILE_W + 2) + (k % 3 - 1);
  }
  // K == 4: (-1,0), (1,0), (0,-1), (0,1)
  return b == 0 ? -(TILE_W + 2) : b == 1 ? (TILE_W + 2) : b == 2 ? -1 : 1;
}

struct TileBlocks {               // how workgroups map to (tile, cell block)
  const BgnnTileMeta *tiles;
  const BgnnWorkItem *items2;     // {tile, r0, c0}; nullptr when every tile has one shape
  int bh, bw;                     // blocks per tile (uniform case)
  int n_blocks;
};

struct BlockPos {
  int tile, r0, c0, h, w;
  int64_t cell_off;
};

// XCD-aware block order: consecutive work items (adjacent cell blocks, which share halo rows) run on
// the same XCD and hit its L2.  Bijective for any n_blocks.
template <int TH = TILE_H>
__device__ __forceinline__ BlockPos decode_block_at(const TileBlocks &tb, int bid);
template <int TH = TILE_H>
__device__ __forceinline__ BlockPos decode_block(const TileBlocks &tb) { return decode_block_at<TH>(tb, blockIdx.x); }
// (explicit index: a persistent workgroup walks bid = blockIdx.x + i * gridDim.x; with a grid that is a multiple of 8 it stays on
//  its XCD's contiguous range of work items and the workgroups of an XCD process adjacent items at the same time)
template <int TH>
__device__ __forceinline__ BlockPos decode_block_at(const TileBlocks &tb, int bid) {
  const int nb = tb.n_blocks;
  const int xcd = bid & 7, q = nb >> 3, r = nb & 7;
  const int wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  BlockPos p;
  if (tb.items2) {
    const BgnnWorkItem it = tb.items2[wid];
    p.tile = it.tile; p.r0 = it.r0; p.c0 = it.nr;
  } else {
    const int bpt = tb.bh * tb.bw;
    p.tile = wid / bpt;
    const int rem = wid - p.tile * bpt;
    p.r0 = (rem / tb.bw) * TH; p.c0 = (rem % tb.bw) * TILE_W;
  }
  const BgnnTileMeta t = tb.tiles[p.tile];
  p.h = t.h; p.w = t.w; p.cell_off = t.cell_off;
  return p;
}

// phase 0: node ids of the 18x18 halo (-1 = outside the tile or invalid) and their alpha_src
template <int H, int NTHREADS, int HROWS = HALO_ROWS>
__device__ __forceinline__ void load_halo_ids(const BlockPos &p, const int32_t *node_id, const float *asd, int *hid,
                                              float *has) {
  for (int idx = threadIdx.x; idx < HROWS; idx += NTHREADS) {
    const int gr = p.r0 + idx / HALO_W - 1, gc = p.c0 + idx % HALO_W - 1;
    int id = -1;
    if (gr >= 0 && gr < p.h && gc >= 0 && gc < p.w) {
      id = node_id[p.cell_off + (int64_t)gr * p.w + gc];
      if (id < 0) id = -1;
    }
    hid[idx] = id;
#pragma unroll
    for (int hh = 0; hh < H; ++hh) has[idx * H + hh] = id >= 0 ? asd[(int64_t)id * 2 * H + hh] : 0.0f;
  }
}

// phase A: attention coefficients of cell `self_idx` (node `my`) for heads [H0, H0+NH):
//   e = leaky_relu(a_src[j] + a_dst[i] + ea . V, 0.2); self loop uses the mean incoming attribute;
//   alpha = exp(e - max) / (sum + 1e-16).   out[b*NH + (hh-H0)], b = 0..K (K = self loop)
template <int H, int K, int H0, int NH>
__device__ __forceinline__ void attention_coefficients(int my, int self_idx, const int *hid, const float *has,
                                                       const float *asd, const float *eattr, const float *V, int ED,
                                                       float *out) {
  using Off = StencilOffsets<K>;
  float ea[K][4];
  float ea_sum[4] = {0.f, 0.f, 0.f, 0.f};
  int deg = 0;
  bool present[K];
  float eraw[K * 4];
  if (ED == 3 && (K * 3) % 4 == 0) {                 // the usual [K][3] block: K*12 bytes, 16-byte aligned rows
    const float4 *ep = reinterpret_cast<const float4 *>(eattr + (int64_t)my * K * 3);
#pragma unroll
    for (int i = 0; i < K * 3 / 4; ++i) {
      const float4 v = ep[i];
      eraw[4 * i] = v.x; eraw[4 * i + 1] = v.y; eraw[4 * i + 2] = v.z; eraw[4 * i + 3] = v.w;
    }
  }
#pragma unroll
  for (int b = 0; b < K; ++b) {
    const int nidx = self_idx - Off::dr[b] * HALO_W - Off::dc[b];   // slot b <- source at -offset[b]
    present[b] = hid[nidx] >= 0;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      float e = 0.0f;
      if (f < ED) e = (ED == 3 && (K * 3) % 4 == 0) ? eraw[(b * 3 + f) & (K * 4 - 1)] : eattr[((int64_t)my * K + b) * ED + f];
      ea[b][f] = (f < ED && present[b]) ? e : 0.0f;
      if (present[b]) ea_sum[f] += ea[b][f];
    }
    deg += present[b] ? 1 : 0;
  }
  const float cnt = (float)(deg > 0 ? deg : 1);      // scatter(..., reduce='mean'): sum / max(count, 1)
#pragma unroll
  for (int k = 0; k < NH; ++k) {
    const int hh = H0 + k;
    float v[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) v[f] = f < ED ? V[hh * ED + f] : 0.0f;
    const float ad = asd[(int64_t)my * 2 * H + H + hh];
    float mx = -__builtin_inff();
    float lg[K + 1];
#pragma unroll
    for (int b = 0; b < K; ++b) {
      const int nidx = self_idx - Off::dr[b] * HALO_W - Off::dc[b];
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
      const float pe = on ? expf(lg[b] - mx) : 0.0f;
      lg[b] = pe;
      den += pe;
    }
    den += 1e-16f;
#pragma unroll
    for (int b = 0; b <= K; ++b) out[b * NH + k] = lg[b] / den;
  }
}

// same as attention_coefficients for ONE head given at run time; out[b], b = 0..K
template <int H, int K>
__device__ __forceinline__ void attention_coefficients_head(int my, int self_idx, int hh, const int *hid, const float *has,
                                                            const float *asd, const float *eattr, const float *V, int ED,
                                                            float *out) {
  using Off = StencilOffsets<K>;
  float v[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) v[f] = f < ED ? V[hh * ED + f] : 0.0f;
  const float ad = asd[(int64_t)my * 2 * H + H + hh];
  float ea_sum[4] = {0.f, 0.f, 0.f, 0.f};
  int deg = 0;
  float mx = -__builtin_inff();
  float lg[K + 1];
  bool present[K];
  const bool vec = (ED == 3 && (K * 3) % 4 == 0);
  float eraw[K * 4];
  if (vec) {
    const float4 *ep = reinterpret_cast<const float4 *>(eattr + (int64_t)my * K * 3);
#pragma unroll
    for (int i = 0; i < K * 3 / 4; ++i) {
      const float4 q = ep[i];
      eraw[4 * i] = q.x; eraw[4 * i + 1] = q.y; eraw[4 * i + 2] = q.z; eraw[4 * i + 3] = q.w;
    }
  }
#pragma unroll
  for (int b = 0; b < K; ++b) {
    const int nidx = self_idx - Off::dr[b] * HALO_W - Off::dc[b];
    present[b] = hid[nidx] >= 0;
    float dot = 0.0f;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      float e = 0.0f;
      if (f < ED && present[b]) e = vec ? eraw[(b * 3 + f) & (K * 4 - 1)] : eattr[((int64_t)my * K + b) * ED + f];
      ea_sum[f] += e;
      dot += e * v[f];
    }
    float x = has[nidx * H + hh] + ad + dot;
    x = x > 0.0f ? x : 0.2f * x;
    lg[b] = x;
    if (present[b]) { mx = fmaxf(mx, x); ++deg; }
  }
  {
    const float cnt = (float)(deg > 0 ? deg : 1);
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
    const float pe = on ? expf(lg[b] - mx) : 0.0f;
    lg[b] = pe;
    den += pe;
  }
  den += 1e-16f;
#pragma unroll
  for (int b = 0; b <= K; ++b) out[b] = lg[b] / den;
}

// ---- attention coefficients of the fused kernels, every operand already in registers -------------------------------------------
// Split in two so that what does not depend on the head is computed ONCE per cell (it used to be redone per head: the operand
// reads are asm, so the compiler could not see that both heads mask the same edge attributes):
//   EdgeTerms: which stencil sources exist, their edge attributes (0 where absent), and the self loop's attributes = mean of the
//              present ones (GATConv fill_value = 'mean'; scatter-mean = sum / max(count, 1));
//   per head : e_b = leaky_relu(a_src[b] + a_dst + ea_b . V, 0.2), softmax over the present sources and the self loop
//              (exp(e - max) / (sum + 1e-16)).
// A lane that owns TWO heads (H = 4: heads hl and hl + 2) runs them as the two halves of packed f32 operations
// (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32): the same IEEE operations in the same order per head, so the pair form and the
// single-head form agree bit for bit (the persistent kernel and the heads instances use the single form; tests/test_gpu_forward.py).
// exp through v_exp_f32 (2^x), one v_rcp_f32 per head instead of K + 1 divisions and one for the mean's 1 / count: ~1 ulp each,
// far inside the 1e-4 bar.  leaky_relu(x, 0.2) = max(x, 0.2 x) for every x (one multiply + one max).
typedef float f32x2_t __attribute__((ext_vector_type(2)));

template <int K>
struct EdgeTerms {
  bool present[K];
  float e[K][3];         // edge attributes of slot b, 0 where the source is absent
  float mean[3];         // the self loop's attributes
};

template <int K>
__device__ __forceinline__ void edge_terms(const int (&nb)[K], const float (&eraw)[K * 3], EdgeTerms<K> &t) {
  float sum[3] = {0.f, 0.f, 0.f};
  int deg = 0;
#pragma unroll
  for (int b = 0; b < K; ++b) {
    t.present[b] = nb[b] >= 0;
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      t.e[b][f] = t.present[b] ? eraw[b * 3 + f] : 0.0f;
      sum[f] += t.e[b][f];
    }
    deg += t.present[b] ? 1 : 0;
  }
  const float rc = __builtin_amdgcn_rcpf((float)(deg > 0 ? deg : 1));
#pragma unroll
  for (int f = 0; f < 3; ++f) t.mean[f] = sum[f] * rc;
}

// the same terms from the COMPACT edge storage (graph_build.hip, FeatureArgs): slot b's attributes are (length of its offset,
// nan_to_num(depth[target] - depth[source]), slope) -- the length is one of the tile's three unit lengths (x, y, diagonal), exactly
// doubled for a dilated slot; the depth difference is the float32 subtraction the feature kernel takes.  Bit for bit the values
// of the full table, in the same summation order.
template <int K>
__device__ __forceinline__ void edge_terms_compact(const int (&nb)[K], const float (&slope)[K], const float (&dsrc)[K + 1], float len_x,
                                                   float len_y, float len_d, EdgeTerms<K> &t) {
  using Off = StencilOffsets<K>;
  float sum[3] = {0.f, 0.f, 0.f};
  int deg = 0;
#pragma unroll
  for (int b = 0; b < K; ++b) {
    constexpr float FMAX = 3.4028234663852886e38f;
    const int adr = Off::dr[b] < 0 ? -Off::dr[b] : Off::dr[b], adc = Off::dc[b] < 0 ? -Off::dc[b] : Off::dc[b];
    const float unit = adc == 0 ? len_y : adr == 0 ? len_x : len_d;
    const float len = (adr > 1 || adc > 1) ? unit + unit : unit;
    const float dz = dsrc[K] - dsrc[b];
    const float dzc = __builtin_fminf(__builtin_fmaxf(dz, -FMAX), FMAX);       // np.nan_to_num: +-inf -> +-FLT_MAX ...
    t.present[b] = nb[b] >= 0;
    t.e[b][0] = t.present[b] ? len : 0.0f;
    t.e[b][1] = t.present[b] ? (dz != dz ? 0.0f : dzc) : 0.0f;                  // ... NaN -> 0
    t.e[b][2] = slope[b];                                                      // (0 where the source is absent: written so)
#pragma unroll
    for (int f = 0; f < 3; ++f) sum[f] += t.e[b][f];
    deg += t.present[b] ? 1 : 0;
  }
  const float rc = __builtin_amdgcn_rcpf((float)(deg > 0 ? deg : 1));
#pragma unroll
  for (int f = 0; f < 3; ++f) t.mean[f] = sum[f] * rc;
}

__device__ __forceinline__ float leaky02(float x) {
  float r;
  const float y = 0.2f * x;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));     // (asm: the C fmax adds a canonicalising instruction)
  return r;
}

// MASKED = false: the caller's alpha_src is -inf where the source is absent (the fused kernel writes its halo table that way), so the
// logit is -inf, drops out of the max and its exp is exactly 0 -- the same values as the masked form, without the selects and without
// keeping the presence flags alive next to the logits.
template <int K, bool MASKED = true>
__device__ __forceinline__ void attention_head(const EdgeTerms<K> &t, const float (&hs)[K + 1], float ad, const float (&v)[3], float *out) {
  float mx = -__builtin_inff();
  float lg[K + 1];
#pragma unroll
  for (int b = 0; b <= K; ++b) {
    const float *e = b < K ? t.e[b < K ? b : 0] : t.mean;
    float dot = e[0] * v[0];
    dot = __builtin_fmaf(e[1], v[1], dot);
    dot = __builtin_fmaf(e[2], v[2], dot);
    const float x = leaky02((hs[b] + ad) + dot);
    lg[b] = x;
    const bool on = b == K || !MASKED ? true : t.present[b < K ? b : 0];
    mx = fmaxf(mx, on ? x : -__builtin_inff());
  }
  float den = 0.0f;
#pragma unroll
  for (int b = 0; b <= K; ++b) {
    const bool on = b == K || !MASKED ? true : t.present[b < K ? b : 0];
    const float pe = on ? __builtin_amdgcn_exp2f((lg[b] - mx) * 1.44269504088896340736f) : 0.0f;
    lg[b] = pe;
    den += pe;
  }
  den += 1e-16f;
  const float rden = __builtin_amdgcn_rcpf(den);
#pragma unroll
  for (int b = 0; b <= K; ++b) out[b] = lg[b] * rden;
}

// two heads of one cell at once: element 0 = the first head, element 1 = the second
template <int K, bool MASKED = true>
__device__ __forceinline__ void attention_head_pair(const EdgeTerms<K> &t, const float (&hs0)[K + 1], const float (&hs1)[K + 1], float ad0,
                                                    float ad1, const float (&v0)[3], const float (&v1)[3], float *out0, float *out1) {
  const f32x2_t ad = {ad0, ad1};
  const f32x2_t w0 = {v0[0], v1[0]}, w1 = {v0[1], v1[1]}, w2 = {v0[2], v1[2]};
  float mx0 = -__builtin_inff(), mx1 = -__builtin_inff();
  f32x2_t lg[K + 1];
#pragma unroll
  for (int b = 0; b <= K; ++b) {
    const float *e = b < K ? t.e[b < K ? b : 0] : t.mean;
    const f32x2_t e0 = {e[0], e[0]}, e1 = {e[1], e[1]}, e2 = {e[2], e[2]};
    f32x2_t dot = e0 * w0;
    dot = __builtin_elementwise_fma(e1, w1, dot);
    dot = __builtin_elementwise_fma(e2, w2, dot);
    const f32x2_t hs = {hs0[b], hs1[b]};
    const f32x2_t xs = (hs + ad) + dot;
    const f32x2_t x = {leaky02(xs.x), leaky02(xs.y)};
    lg[b] = x;
    const bool on = b == K || !MASKED ? true : t.present[b < K ? b : 0];
    mx0 = fmaxf(mx0, on ? x.x : -__builtin_inff());
    mx1 = fmaxf(mx1, on ? x.y : -__builtin_inff());
  }
  const f32x2_t mx = {mx0, mx1};
  const f32x2_t l2e = {1.44269504088896340736f, 1.44269504088896340736f};
  f32x2_t den = {0.0f, 0.0f};
#pragma unroll
  for (int b = 0; b <= K; ++b) {
    const bool on = b == K || !MASKED ? true : t.present[b < K ? b : 0];
    const f32x2_t arg = (lg[b] - mx) * l2e;
    f32x2_t pe = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
    if (!on) pe = (f32x2_t){0.0f, 0.0f};
    lg[b] = pe;
    den += pe;
  }
  den += (f32x2_t){1e-16f, 1e-16f};
  const f32x2_t rden = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
#pragma unroll
  for (int b = 0; b <= K; ++b) {
    const f32x2_t o = lg[b] * rden;
    out0[b] = o.x; out1[b] = o.y;
  }
}

// one head, from the raw operands (`nb` = node ids of the K stencil sources (< 0: absent), `hs` = alpha_src of the K sources and
// (slot K) of the node itself, `eraw` = the node's [K][3] edge attributes, `ad` = its alpha_dst, `v` = V[head][0..2])
template <int K>
__device__ __forceinline__ void attention_coefficients_head_vals(const int (&nb)[K], const float (&hs)[K + 1],
                                                                 const float (&eraw)[K * 3], float ad, const float (&v)[3],
                                                                 float *out) {
  EdgeTerms<K> t;
  edge_terms<K>(nb, eraw, t);
  attention_head<K>(t, hs, ad, v, out);
}

// attention_coefficients_head with the node's own operands already in registers (ED == 3): `eraw` = its [K][3]
// edge-attribute block, `ad` = its alpha_dst for head hh, `v` = V[hh][0..2].  Lets the caller issue those global loads before the
// halo ids are known (one latency less on the workgroup's critical path).
template <int H, int K, int HWID = HALO_W>
__device__ __forceinline__ void attention_coefficients_head_pre(int self_idx, int hh, const int *hid, const float *has,
                                                                const float (&eraw)[K * 3], float ad, const float (&v)[3],
                                                                float *out) {
  using Off = StencilOffsets<K>;
  int nb[K];
  float hs[K + 1];
#pragma unroll
  for (int b = 0; b < K; ++b) {
    const int nidx = self_idx - Off::dr[b] * HWID - Off::dc[b];
    nb[b] = hid[nidx];
    hs[b] = has[nidx * H + hh];
  }
  hs[K] = has[self_idx * H + hh];
  attention_coefficients_head_vals<K>(nb, hs, eraw, ad, v, out);
}

}  // namespace bgnn
