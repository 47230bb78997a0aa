// Fused GAT layer for stencil graphs (gfx950):  aggregate_l  ->  BN/ReLU  ->  GEMM_{l+1}
//
// One workgroup (256 threads = 4 waves, one per SIMD, one workgroup per CU) owns a 16x16 block of cells of one tile and
// produces, for those 256 nodes, what the NEXT stage needs:
//   EPI_NEXT : xw_{l+1} = h_{l+1} @ W_{l+1}^T  and its attention dots (alpha_src, alpha_dst)
//   EPI_HEADS: the three output heads, softmax / argmax / sigmoid, the predict() flags and
//              (optionally) the classification / confidence / correction GRIDS of
//              BathymetricPipeline._process_tile -- i.e. K4(last) + K5 + K6 in one launch.
// h_{l+1} (the aggregate output, 1 KiB/node) never goes to HBM: per 32-channel slab it is
// produced by the LDS-tiled gather (as gat_aggregate_tiled.hip), staged in LDS and immediately
// consumed as the B operand of a rank-32 MFMA update of the block's [256 nodes x NC] accumulator.
// HBM traffic per layer drops from read xw + write h + read h + write xw' (4.3 KiB/node at
// HC = 256) to read xw (x1.27 halo) + write xw' (2.3 KiB/node); the kernel is then bound by the
// exact-f32 matrix pipe.
//
// Per slab s:   [halo slab regs -> LDS] | barrier | gather+epilogue -> stage | barrier |
//               issue next halo loads | 2 x 32 x NT MFMAs per wave | barrier | next W chunk (LDS-DMA)
// so the halo loads of slab s+1 hide under the MFMAs of slab s and the W chunk under its gather.
//
// Reference semantics: models/gnn.py:173-188 (conv -> norm -> relu), :392-406 (heads),
// :427-449 (predict), models/pipeline.py:278-307 (grids); GATConv per SURVEY Appendix B.
#include <stdlib.h>
#include "gat_tile_common.h"

namespace bgnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { EPI_NEXT = 0, EPI_HEADS = 1 };

struct FusedArgs {
  TileBlocks tb;
  const int32_t *node_id;
  const float *xw;        // [rows][HC]   this layer's lin(x)
  const float *asd;       // [rows][2H]
  const float *eattr;     // [rows][K][ED]
  const float *V;         // [H][ED]
  const float *scale;     // [HC] folded bias + BatchNorm
  const float *shift;
  const float *Wt;        // [HC][NC] next stage weight (transposed)
  const float *zero_page; // >= 16 B of zeros (source of halo rows that have no node)
  int ED, relu, dbg;
  // EPI_NEXT
  const float *att_src;   // [NC]
  const float *att_dst;
  float *out;             // [rows][NC]
  float *asd_out;         // [rows][2*H2]
  int H2, C2;
  // EPI_HEADS
  const float *hd_b0;     // [NC]
  const float *hd_W1;     // cls [classes][hh], conf [hh], corr [hh]
  const float *hd_b1;
  const float *local_std; // [rows]
  int classes, hh, has_corr;
  float thr_auto, thr_review, norm_floor;
  bgnn_outputs o;
  float *cls_grid, *conf_grid, *corr_grid;   // [cells] or nullptr
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS reads issued from inline asm.  hipcc cannot tell an LDS-DMA's destination from the address of a
// later ds_read, so with a DMA in flight it puts s_waitcnt vmcnt(0) in front of every compiler-visible
// LDS read -- which would serialise the slab / W prefetch with the phase it is meant to hide under.
// These reads are invisible to that pass; the code below waits for them explicitly (lgkmcnt) and fences
// the scheduler (sched_barrier) as the HIP guide's rule 18 requires.
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
template <int OFF>
__device__ __forceinline__ f32x4 lds_read4(uint32_t addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ float lds_read1(uint32_t addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ void lds_reads_done() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <int N>
__device__ __forceinline__ void wait_vm_lgkm() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int NT>
__device__ __forceinline__ void stage_w_chunk(const float *Wt, float *dst, int k0, int wave, int lane) {
  // rows k0..k0+31 of Wt[.][NC]: 32*NC contiguous floats = NT*4 pieces of 1 KiB, 4 waves
  constexpr int NC = NT * 32;
  const char *src = reinterpret_cast<const char *>(Wt + (int64_t)k0 * NC);
  constexpr int NQ = NT * 4;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int q = j * 4 + wave;
    if (q < NQ)
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const void *>(src + q * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void *)(dst + q * 256), 16, 0, 0);
  }
}

template <int HC, int C, int K, int NT, int EPI>
__global__ __launch_bounds__(256, 1) void gat_layer_fused_kernel(FusedArgs a) {
  // 4 waves, ONE per SIMD, each with the whole 512-register budget: a wave's accumulator is
  // 64 nodes x NC channels (2 row groups x NT tiles x 16 regs = 256 registers at NC = 256).
  constexpr int H = HC / C;
  constexpr int NC = NT * 32;
  constexpr int NSLAB = HC / 32, SPH = C / 32;
  constexpr int HR = HALO_ROWS, HW_ = HALO_W;
  using Off = StencilOffsets<K>;
  extern __shared__ __attribute__((aligned(128))) float lds[];
  float *slab = lds;                                   // [2][HR][32]  halo rows of slab s / s+1, 16-B chunks XOR-swizzled
  float *wbuf = slab + 2 * HR * 32;                    // [2][32][NC]  W_{l+1} rows of slab s / s+1
  float *scsh = wbuf + 2 * 32 * NC;                    // [2][HC]      folded scale / shift
  int *hid = reinterpret_cast<int *>(scsh + 2 * HC);   // [HR]
  float *has = reinterpret_cast<float *>(hid + HR);    // [HR][H]

  const BlockPos pos = decode_block(a.tb);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tr = tid / TILE_W, tc = tid % TILE_W;      // thread = cell in the gather phase
  const int self_idx = (tr + 1) * HW_ + tc + 1;

  load_halo_ids<H, 256>(pos, a.node_id, a.asd, hid, has);
  for (int i = tid; i < HC; i += 256) { scsh[i] = a.scale[i]; scsh[HC + i] = a.shift[i]; }
  __syncthreads();

  // Halo rows go global -> LDS by LDS-DMA (no VGPR staging, nothing live across the MFMA phase).  A
  // wave-instruction writes 64 x 16 B linearly = 8 rows x 128 B, so rows are unpadded and the bank
  // spreading is an XOR swizzle applied on the SOURCE side: LDS chunk p of a row holds channel chunk
  // p ^ ((row >> 1) & 7).  Rows that have no node read a zero page.
  constexpr int NPIECE = (HR * 8 + 255) / 256;
  auto issue_slab = [&](int s, float *dst) {
#pragma unroll 1                                         // (unrolled, hipcc hoists + spills the addresses and
    for (int j = 0; j < NPIECE; ++j) {                   //  every reload's vmcnt(0) serialises the DMAs)
      const int idx = j * 256 + tid;
      if (idx < HR * 8) {
        const int row = idx >> 3, c = (idx & 7) ^ ((row >> 1) & 7);
        const int id = hid[row];
        const float *src = id >= 0 ? a.xw + (int64_t)id * HC + s * 32 + c * 4 : a.zero_page;
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const void *>(src),
                                         (__attribute__((address_space(3))) void *)(dst + (j * 256 + wave * 64) * 4), 16, 0, 0);
      }
    }
  };
  issue_slab(0, slab);
  stage_w_chunk<NT>(a.Wt, wbuf, 0, wave, lane);

  // ---- phase A: attention coefficients of this thread's cell, all heads, kept in registers ----------
  float alf[(K + 1) * H];
  {
    const int my = hid[self_idx];
#pragma unroll
    for (int i = 0; i < (K + 1) * H; ++i) alf[i] = 0.0f;
    if (my >= 0) attention_coefficients<H, K, 0, H>(my, self_idx, hid, has, a.asd, a.eattr, a.V, a.ED, alf);
  }

  f32x16 acc[2][NT];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[g][t][i] = 0.0f;

  const int r = lane & 31, hl = lane >> 5;              // MFMA lane roles: node r of a row group, k-half hl
  // neighbour rows of this cell in the swizzled slab image: byte offset of chunk 0's slot
  const uint32_t slab0 = lds_addr(slab);
  const uint32_t wbuf0 = lds_addr(wbuf + 4 * hl * NC + r);
  const uint32_t scsh0 = lds_addr(scsh);

  // ---- slabs: ONE barrier each.  Slab s+1 and W chunk s+1 stream into the other buffers while slab s is
  // gathered and multiplied, so their latency is hidden; the gather output never touches LDS: the wave that
  // gathers cells 64w..64w+63 is the wave that multiplies them, and v_permlane32_swap puts each value in the
  // lane half the MFMA B operand wants.
#pragma unroll
  for (int hh = 0; hh < H; ++hh) {
#pragma unroll 1
    for (int ss = 0; ss < SPH; ++ss) {
      const int s = hh * SPH + ss;
      const int buf = s & 1;
      wait_vm_lgkm<0>();
      __builtin_amdgcn_s_barrier();                     // slab s / W s visible; everyone is past iteration s-1
      if (s + 1 < NSLAB) {
        issue_slab(s + 1, slab + (buf ^ 1) * HR * 32);
        stage_w_chunk<NT>(a.Wt, wbuf + (buf ^ 1) * 32 * NC, (s + 1) * 32, wave, lane);
      }
      // gather: g[q] = sum_b alpha[b] * row_b[chunk q]
      f32x4 g[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) g[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const uint32_t sb = slab0 + buf * (HR * 128);
#pragma unroll
      for (int b = 0; b <= K; ++b) {
        const int nidx = b == K ? self_idx : self_idx - Off::dr[b < K ? b : 0] * HW_ - Off::dc[b < K ? b : 0];
        const uint32_t rb = sb + nidx * 128 + (((nidx >> 1) & 7) << 4);   // slot of channel chunk 0
        f32x4 x[8];
        x[0] = lds_read4<0>(rb); x[1] = lds_read4<0>(rb ^ 16); x[2] = lds_read4<0>(rb ^ 32); x[3] = lds_read4<0>(rb ^ 48);
        x[4] = lds_read4<0>(rb ^ 64); x[5] = lds_read4<0>(rb ^ 80); x[6] = lds_read4<0>(rb ^ 96); x[7] = lds_read4<0>(rb ^ 112);
        lds_reads_done();
        const float alpha = alf[b * H + hh];
#pragma unroll
        for (int q = 0; q < 8; ++q) g[q] += alpha * x[q];
      }
      // layer epilogue: (+bias, BatchNorm) folded, ReLU -> h_{l+1}[cell][s*32 ..], in registers
      {
        const uint32_t cp = scsh0 + s * 128;
        f32x4 sc[8], sh[8];
        sc[0] = lds_read4<0>(cp); sc[1] = lds_read4<16>(cp); sc[2] = lds_read4<32>(cp); sc[3] = lds_read4<48>(cp);
        sc[4] = lds_read4<64>(cp); sc[5] = lds_read4<80>(cp); sc[6] = lds_read4<96>(cp); sc[7] = lds_read4<112>(cp);
        sh[0] = lds_read4<HC * 4>(cp); sh[1] = lds_read4<HC * 4 + 16>(cp); sh[2] = lds_read4<HC * 4 + 32>(cp);
        sh[3] = lds_read4<HC * 4 + 48>(cp); sh[4] = lds_read4<HC * 4 + 64>(cp); sh[5] = lds_read4<HC * 4 + 80>(cp);
        sh[6] = lds_read4<HC * 4 + 96>(cp); sh[7] = lds_read4<HC * 4 + 112>(cp);
        lds_reads_done();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          g[q] = g[q] * sc[q] + sh[q];
          if (a.relu) {
            g[q].x = g[q].x > 0.f ? g[q].x : 0.f; g[q].y = g[q].y > 0.f ? g[q].y : 0.f;
            g[q].z = g[q].z > 0.f ? g[q].z : 0.f; g[q].w = g[q].w > 0.f ? g[q].w : 0.f;
          }
        }
      }
      // B operands: MFMA k-step (s8, i) takes channel 8*s8 + 4*hl + i of node r (row group 0) and node r+32
      // (row group 1).  Lane L holds every channel of node L: chunks 2*s8 (hl = 0) and 2*s8+1 (hl = 1).
      // permlane32_swap(A, B) -> A' = [low lanes: A.low | high lanes: B.low], B' = [A.high | B.high]
      f32x4 x0[4], x1[4];
#pragma unroll
      for (int s8 = 0; s8 < 4; ++s8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float lo = g[2 * s8][i], hi = g[2 * s8 + 1][i];
          // lanes 32-63 of `lo` <-> lanes 0-31 of `hi`
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
          x0[s8][i] = lo;
          x1[s8][i] = hi;
        }
      }
      // rank-32 update of the block's accumulator
      const uint32_t wa = wbuf0 + buf * (32 * NC * 4);
#pragma unroll
      for (int s8 = 0; s8 < 4; ++s8) {
        float wv[4][NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          wv[0][t] = lds_read1<(0) * 4>(wa + (s8 * 8 * NC + t * 32) * 4);
          wv[1][t] = lds_read1<(NC) * 4>(wa + (s8 * 8 * NC + t * 32) * 4);
          wv[2][t] = lds_read1<(2 * NC) * 4>(wa + (s8 * 8 * NC + t * 32) * 4);
          wv[3][t] = lds_read1<(3 * NC) * 4>(wa + (s8 * 8 * NC + t * 32) * 4);
        }
        lds_reads_done();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[i][t], x0[s8][i], acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[i][t], x1[s8][i], acc[1][t], 0, 0, 0);
          }
        }
      }
    }
  }
  const int mrow0 = wave * 64 + r;                      // block-local node (cell) of row group 0; group 1 = +32

  // ---- epilogue: lane (r, hl) holds node mrow; reg i of tile t -> channel t*32 + 8*(i>>2) + 4*hl + (i&3) ----
#pragma unroll
  for (int rg = 0; rg < 2; ++rg) {
    const int mrow = mrow0 + 32 * rg;
    const int mr = mrow / TILE_W, mc = mrow % TILE_W;
    const int id = hid[(mr + 1) * HW_ + mc + 1];
    if (EPI == EPI_NEXT) {
      // next layer's attention dots (its width per head is C as well): tile t belongs to head t / (C/32)
      constexpr int TPH = C / 32, H2 = NT / TPH;
      float ps[H2], pd[H2];
#pragma unroll
      for (int hd = 0; hd < H2; ++hd) { ps[hd] = 0.0f; pd[hd] = 0.0f; }
      float *yp = a.out + (int64_t)(id >= 0 ? id : 0) * NC + 4 * hl;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 s4 = *reinterpret_cast<const float4 *>(a.att_src + t * 32 + 8 * g + 4 * hl);
          const float4 d4 = *reinterpret_cast<const float4 *>(a.att_dst + t * 32 + 8 * g + 4 * hl);
          const float4 v = make_float4(acc[rg][t][4 * g], acc[rg][t][4 * g + 1], acc[rg][t][4 * g + 2], acc[rg][t][4 * g + 3]);
          ps[t / TPH] += v.x * s4.x + v.y * s4.y + v.z * s4.z + v.w * s4.w;
          pd[t / TPH] += v.x * d4.x + v.y * d4.y + v.z * d4.z + v.w * d4.w;
          if (id >= 0) *reinterpret_cast<float4 *>(yp + t * 32 + 8 * g) = v;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int hd = 0; hd < H2; ++hd) {
        const float s_ = ps[hd] + __shfl_xor(ps[hd], 32);
        const float d_ = pd[hd] + __shfl_xor(pd[hd], 32);
        if (id >= 0 && hl == 0) {
          a.asd_out[(int64_t)id * 2 * H2 + hd] = s_;
          a.asd_out[(int64_t)id * 2 * H2 + H2 + hd] = d_;
        }
      }
    } else {
      // heads: hidden = relu(acc + b0); with hidden/2 == 32 tile 0 = classification, 1 = confidence,
      // 2 = correction hidden units.  Second layers: in-lane partial dots + one cross-half add.
      static_assert(EPI == EPI_NEXT || C == 64, "heads epilogue assumes hidden/2 == 32 (one accumulator tile per head)");
      const int ncls = a.classes;
      float lg[4] = {0.f, 0.f, 0.f, 0.f};
      float sc = 0.0f, sr = 0.0f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c0 = 8 * g + 4 * hl;                 // unit index within the head
          const float4 b4 = *reinterpret_cast<const float4 *>(a.hd_b0 + t * 32 + c0);
          float v[4] = {acc[rg][t][4 * g] + b4.x, acc[rg][t][4 * g + 1] + b4.y, acc[rg][t][4 * g + 2] + b4.z,
                        acc[rg][t][4 * g + 3] + b4.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
          if (t == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if (k < ncls) {
                const float4 w4 = *reinterpret_cast<const float4 *>(a.hd_W1 + k * 32 + c0);
                lg[k] += v[0] * w4.x + v[1] * w4.y + v[2] * w4.z + v[3] * w4.w;
              }
            }
          } else if (t == 1) {
            const float4 w4 = *reinterpret_cast<const float4 *>(a.hd_W1 + ncls * 32 + c0);
            sc += v[0] * w4.x + v[1] * w4.y + v[2] * w4.z + v[3] * w4.w;
          } else if (a.has_corr) {
            const float4 w4 = *reinterpret_cast<const float4 *>(a.hd_W1 + (ncls + 1) * 32 + c0);
            sr += v[0] * w4.x + v[1] * w4.y + v[2] * w4.z + v[3] * w4.w;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) lg[k] += __shfl_xor(lg[k], 32);
      sc += __shfl_xor(sc, 32);
      sr += __shfl_xor(sr, 32);
      const int gr = pos.r0 + mr, gc = pos.c0 + mc;
      const bool inside = gr < pos.h && gc < pos.w;
      if (hl == 0 && inside) {
        const int64_t cidx = pos.cell_off + (int64_t)gr * pos.w + gc;
        float fcls = 0.f, fconf = 0.f, fcorr = 0.f;
        if (id >= 0) {
          float mx = -__builtin_inff();
#pragma unroll
          for (int k = 0; k < 4; ++k) if (k < ncls) { lg[k] += a.hd_b1[k]; mx = fmaxf(mx, lg[k]); }
          float pr[4], den = 0.0f;
#pragma unroll
          for (int k = 0; k < 4; ++k) { pr[k] = k < ncls ? expf(lg[k] - mx) : 0.0f; den += pr[k]; }
          int arg = 0;
          float best = -1.0f;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            pr[k] = pr[k] / den;
            if (k < ncls && pr[k] > best) { best = pr[k]; arg = k; }
          }
          const float conf = 1.0f / (1.0f + expf(-(sc + a.hd_b1[ncls])));
          const float corr = sr + a.hd_b1[ncls + 1];
          const int64_t n = id;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (k < ncls) {
              if (a.o.class_logits) a.o.class_logits[n * ncls + k] = lg[k];
              if (a.o.class_probs) a.o.class_probs[n * ncls + k] = pr[k];
            }
          }
          if (a.o.predicted_class) a.o.predicted_class[n] = arg;
          if (a.o.confidence) a.o.confidence[n] = conf;
          if (a.o.correction && a.has_corr) a.o.correction[n] = corr;
          int action = 0;
          if (arg == 2 && conf > a.thr_auto) action = 1;
          if (conf < a.thr_review) action = 2;
          if (a.o.action) a.o.action[n] = action;
          if (a.o.needs_review) a.o.needs_review[n] = action == 2;
          if (a.o.auto_correct) a.o.auto_correct[n] = action == 1;
          fcls = (float)arg; fconf = conf;
          if (a.has_corr && a.corr_grid) {
            float sd = a.local_std[n];
            sd = sd > a.norm_floor ? sd : a.norm_floor;
            fcorr = corr * sd;
          }
        }
        if (a.cls_grid) a.cls_grid[cidx] = fcls;
        if (a.conf_grid) a.conf_grid[cidx] = fconf;
        if (a.corr_grid) a.corr_grid[cidx] = fcorr;
      }
    }
  }
}

template <int HC, int C, int K, int NT, int EPI>
static int launch_inst(bgnn_ctx *ctx, const FusedArgs &a) {
  constexpr int H = HC / C;
  constexpr size_t lds_bytes = (size_t)(2 * HALO_ROWS * 32 + 2 * 32 * NT * 32 + 2 * HC + HALO_ROWS + HALO_ROWS * H) * 4;
  static bool configured = false;     // per instantiation
  auto kern = gat_layer_fused_kernel<HC, C, K, NT, EPI>;
  if (!configured) {
    BGNN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_bytes));
    configured = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.tb.n_blocks), dim3(256), lds_bytes, ctx->stream, a);
  BGNN_HIP_CHECK(hipGetLastError());
  return BGNN_OK;
}

static bool fused_supported(const bgnn_graph *g, int C) {
  return g->kind == 0 && (g->K == 4 || g->K == 8) && g->n_blocks2 > 0 && C == 64;
}

static void fill_common(FusedArgs &a, const bgnn_graph *g, const BgnnLayer &L, int ED, const float *xw, const float *asd,
                        int relu) {
  a.tb.tiles = g->d_tiles; a.tb.items2 = g->uni_h ? nullptr : g->d_items2;
  a.tb.bh = g->bh2; a.tb.bw = g->bw2; a.tb.n_blocks = g->n_blocks2;
  a.node_id = g->d_node_id; a.xw = xw; a.asd = asd; a.eattr = g->d_eattr; a.V = L.V; a.scale = L.scale; a.shift = L.shift;
  a.ED = ED; a.relu = relu; a.zero_page = g->ctx->zero_page;
  { const char *e = getenv("BGNN_FUSED_DBG"); a.dbg = e ? atoi(e) : 0; }
}

// aggregate of layer L (width HC = L.heads*C) fused with the GEMM of the next layer `Ln`
int launch_fused_layer_next(bgnn_ctx *ctx, const bgnn_graph *g, const BgnnLayer &L, const BgnnLayer &Ln, int C, int ED,
                            const float *xw, const float *asd, float *xw_next, float *asd_next) {
  if (!fused_supported(g, C)) return BGNN_ERR_UNSUPPORTED;
  const int HC = L.heads * C, NC = Ln.heads * C;
  if (Ln.d_in != HC) return BGNN_ERR_UNSUPPORTED;
  FusedArgs a{};
  fill_common(a, g, L, ED, xw, asd, L.concat ? 1 : 0);
  a.Wt = Ln.Wt; a.att_src = Ln.att_src; a.att_dst = Ln.att_dst; a.out = xw_next; a.asd_out = asd_next;
  a.H2 = Ln.heads; a.C2 = C;
  ProfScope ps(ctx, BGNN_K_FUSED);
#define BGNN_FUSED_CASE(hc, nt)                                                                         \
  if (HC == hc && NC == nt * 32)                                                                        \
    return g->K == 8 ? launch_inst<hc, 64, 8, nt, EPI_NEXT>(ctx, a) : launch_inst<hc, 64, 4, nt, EPI_NEXT>(ctx, a);
  BGNN_FUSED_CASE(256, 8) BGNN_FUSED_CASE(256, 2) BGNN_FUSED_CASE(128, 4) BGNN_FUSED_CASE(128, 2) BGNN_FUSED_CASE(64, 2)
#undef BGNN_FUSED_CASE
  return BGNN_ERR_UNSUPPORTED;
}

// aggregate of the LAST layer (HC = C, one head) fused with the heads (+ grids)
int launch_fused_layer_heads(bgnn_ctx *ctx, const bgnn_graph *g, const bgnn_model *m, const BgnnLayer &L, int C, int ED,
                             const float *xw, const float *asd, float thr_auto, float thr_review, float norm_floor,
                             const bgnn_outputs *o, float *cls_grid, float *conf_grid, float *corr_grid) {
  if (!fused_supported(g, C) || L.heads != 1 || m->desc.num_classes > 4 || m->head_hidden_total != 96 ||
      (m->desc.predict_correction ? 3 : 2) * (C / 2) > 96 || o->hidden)
    return BGNN_ERR_UNSUPPORTED;
  FusedArgs a{};
  fill_common(a, g, L, ED, xw, asd, L.concat ? 1 : 0);
  a.Wt = m->hd_W0t; a.hd_b0 = m->hd_b0; a.hd_W1 = m->hd_W1; a.hd_b1 = m->hd_b1; a.local_std = g->d_local_std;
  a.classes = m->desc.num_classes; a.hh = C / 2; a.has_corr = m->desc.predict_correction;
  a.thr_auto = thr_auto; a.thr_review = thr_review; a.norm_floor = norm_floor; a.o = *o;
  a.cls_grid = cls_grid; a.conf_grid = conf_grid; a.corr_grid = corr_grid;
  ProfScope ps(ctx, BGNN_K_FUSED);
  return g->K == 8 ? launch_inst<64, 64, 8, 3, EPI_HEADS>(ctx, a) : launch_inst<64, 64, 4, 3, EPI_HEADS>(ctx, a);
}

}  // namespace bgnn
