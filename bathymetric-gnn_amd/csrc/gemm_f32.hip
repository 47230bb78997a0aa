// K3: dense node-feature x weight GEMM in exact float32 on the matrix cores (gfx950).
//
//   Y[M, NC] = act( X[M, K] @ Wt[K, NC] + bias )          (+ fused attention dot products)
//
// Replaces the `lin(x)` GEMM of torch_geometric GATConv and its alpha_src / alpha_dst reductions
// (reference models/gnn.py:176) and the Linear layers of the feature
// extractor / heads (models/gnn.py:52-68, 203-208).  v_mfma_f32_32x32x2_f32 is bit-for-bit a
// k-ordered fmaf chain, so results stay within float32 rounding of the reference's CPU sgemm.
//
// Decomposition: one wave owns 32 rows x all NC columns (NT = NC/32 accumulator tiles); a
// 256-thread workgroup = 4 waves = 128 rows, two workgroups per CU.
//   * X operand: one f32 per lane (lane l: row l&31, k-slot l>>5).  Each lane reads 16 contiguous
//     bytes of ITS OWN row straight from global memory; the two lane halves take k = 4h..4h+3 of
//     every 8-wide k-step, so X never needs an LDS transpose.  Next chunk prefetched in registers.
//   * W operand: the [32 k][NC] chunk of Wt is contiguous in memory, so it is streamed into a
//     double-buffered LDS image by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip) while
//     the previous chunk is being multiplied; fragment reads are conflict-free ds_read_b32.
//   * The product is computed TRANSPOSED (W fragment as the A operand, X fragment as B): each lane
//     then holds 16 output CHANNELS of ONE row (4 groups of 4 consecutive channels), so bias /
//     ReLU / the attention dot products are in-lane work and stores are float4.
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "bgnn_internal.h"

namespace bgnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GEMM_KC = 32;   // k-chunk staged in LDS

struct GemmArgs {
  const float *X;
  const float *Wt;
  const float *bias;
  float *Y;
  const int64_t *d_m;
  const float *att_src;   // [NC] or nullptr
  const float *att_dst;   // [NC]
  float *asd;             // [M][2H]
  float *dump;            // >= 1 KiB scratch row for rows >= M (branch-free stores)
  int ldx, ldy, K, relu, H, C;
  int dbg;                // diagnostic ablations (BGNN_GEMM_DBG): 1 = no row stores, 2 = no MFMAs, 4 = no X loads
  // W-resident form with the feature extractor's first Linear in front (FRONT): X is then the [M][8] node feature table and
  const float *W0t;       // [8][64]  extractor layer 1 (transposed)
  const float *b0;        // [64]
  // one 256-column block of a wider layer (generic kernel only): the attention dots of this block's heads go to columns
  // asd_hd0 .. of the [M][2 asd_H] table (0 / 0: the whole layer in one launch, asd_H = H)
  int asd_H, asd_hd0;
  float w_inv;            // SP = 2: the float16 weight image holds W * 2^S; the accumulators are multiplied by 2^-S
};

template <int NT>
__device__ __forceinline__ void stage_w(const float *Wt, float *dst, int kc, int kn, int wave, int lane) {
  // chunk = rows kc..kc+kn-1 of Wt: kn*NC contiguous floats; one wave-instruction moves 1 KiB
  constexpr int NC = NT * 32;
  const char *src = reinterpret_cast<const char *>(Wt + (int64_t)kc * NC);
  const int bytes = kn * NC * 4;
  constexpr int NQ = GEMM_KC * NC * 4 / 1024;       // 1-KiB pieces in a full chunk
#pragma unroll
  for (int j = 0; j < NQ / 4; ++j) {
    const int q = j * 4 + wave;
    if (q * 1024 + lane * 16 < bytes)
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const void *>(src + q * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void *)(dst + q * 256), 16, 0, 0);
  }
}

template <int NT, bool ATT>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs a) {
  constexpr int NC = NT * 32;
  constexpr int WL = 2 * GEMM_KC * NC > 4 * 32 * 36 + 3 * NC ? 2 * GEMM_KC * NC : 4 * 32 * 36 + 3 * NC;   // also holds the store patches + att + bias
  __shared__ float wl_[WL];
  float (*wl)[GEMM_KC * NC] = reinterpret_cast<float (*)[GEMM_KC * NC]>(wl_);
  const int64_t M = *a.d_m;
  const int64_t row_block = (int64_t)blockIdx.x * 128;
  if (row_block >= M) return;                       // uniform per block
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int64_t row = row_block + wave * 32 + r;
  const int64_t row_ld = row < M ? row : M - 1;     // clamp loads, mask stores
  const float *xp = a.X + row_ld * a.ldx + 4 * h;
  const int K = a.K;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

  float4 a_cur[GEMM_KC / 8], a_nxt[GEMM_KC / 8];
  {
    const int kn = K < GEMM_KC ? K : GEMM_KC;
    stage_w<NT>(a.Wt, wl[0], 0, kn, wave, lane);
#pragma unroll
    for (int s = 0; s < GEMM_KC / 8; ++s)
      if (s * 8 < kn) a_cur[s] = *reinterpret_cast<const float4 *>(xp + s * 8);
  }
  __syncthreads();                                  // (emits vmcnt(0): chunk 0 has landed)

  int buf = 0;
  for (int kc = 0; kc < K; kc += GEMM_KC) {
    const int kn = (K - kc) < GEMM_KC ? (K - kc) : GEMM_KC;   // multiple of 8
    const int kc2 = kc + GEMM_KC;
    if (kc2 < K) {
      const int kn2 = (K - kc2) < GEMM_KC ? (K - kc2) : GEMM_KC;
      stage_w<NT>(a.Wt, wl[buf ^ 1], kc2, kn2, wave, lane);
#pragma unroll
      for (int s = 0; s < GEMM_KC / 8; ++s)
        if (s * 8 < kn2) a_nxt[s] = *reinterpret_cast<const float4 *>(xp + kc2 + s * 8);
    }
    const float *wb = wl[buf];
#pragma unroll
    for (int s = 0; s < GEMM_KC / 8; ++s) {
      if (s * 8 < kn) {
        const float av[4] = {a_cur[s].x, a_cur[s].y, a_cur[s].z, a_cur[s].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float *wrow = wb + (s * 8 + 4 * h + i) * NC + r;
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[t * 32], av[i], acc[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();        // all waves done with wl[buf]; vmcnt(0): next chunk + a_nxt have landed
#pragma unroll
    for (int s = 0; s < GEMM_KC / 8; ++s) a_cur[s] = a_nxt[s];
    buf ^= 1;
  }

  // epilogue.  Transposed C/D layout: lane (r, h) holds row r; reg i -> channel
  // t*32 + 8*(i>>2) + 4*h + (i&3)
  // Row-per-lane stores are store-issue bound (32 rows x 32 B per instruction).  Each 32x32 tile goes through
  // a wave-private LDS patch (wl is free now) and leaves as whole 128-byte row segments, 8 rows per instruction.
  // With ATT the attention dot products of the same tile are taken on the way (per-tile partials, folded into
  // heads at the end); att_src | att_dst sit in LDS behind the patches.
  __syncthreads();                                   // every wave is past its last read of wl
  float *attl = wl_ + 4 * 32 * 36;
  float *biasl = attl + 2 * NC;                      // the bias too: a global load per tile would queue behind the row stores (vmcnt is in order)
  if (ATT || a.bias) {
    if (ATT)
      for (int i = threadIdx.x; i < NC; i += 256) { attl[i] = a.att_src[i]; attl[NC + i] = a.att_dst[i]; }
    if (a.bias)
      for (int i = threadIdx.x; i < NC; i += 256) biasl[i] = a.bias[i];
    __syncthreads();
  }
  float pts[ATT ? NT : 1], ptd[ATT ? NT : 1];
  {
    float *patch = wl_ + wave * (32 * 36);
    const int64_t wrow0 = row_block + wave * 32;
    float *dst[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rr = (lane >> 3) + 8 * k;
      dst[k] = (wrow0 + rr < M ? a.Y + (wrow0 + rr) * a.ldy : a.dump) + (lane & 7) * 4;
    }
    const uint32_t asl = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)(attl + 4 * h);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float4 v[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        v[g] = make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
        if (a.bias) {
          const float4 b = *reinterpret_cast<const float4 *>(biasl + t * 32 + 8 * g + 4 * h);
          v[g].x += b.x; v[g].y += b.y; v[g].z += b.z; v[g].w += b.w;
        }
        if (a.relu) {
          v[g].x = v[g].x > 0.f ? v[g].x : 0.f; v[g].y = v[g].y > 0.f ? v[g].y : 0.f;
          v[g].z = v[g].z > 0.f ? v[g].z : 0.f; v[g].w = v[g].w > 0.f ? v[g].w : 0.f;
        }
      }
      if (ATT) {
        // asm reads + an explicit wait per tile: left to the scheduler the 8*NT att reads are hoisted above the
        // stores and, next to the live accumulators, spilled -- and a scratch reload waits for the stores in flight
        f32x4 s4[4], d4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          asm volatile("ds_read_b128 %0, %1" : "=v"(s4[g]) : "v"(asl + (t * 32 + 8 * g) * 4));
          asm volatile("ds_read_b128 %0, %1" : "=v"(d4[g]) : "v"(asl + (NC + t * 32 + 8 * g) * 4));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 ps2 = {0.f, 0.f}, pd2 = {0.f, 0.f};       // two-lane partial sums: v_pk_fma_f32
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x2 vlo = {v[g].x, v[g].y}, vhi = {v[g].z, v[g].w};
          ps2 += vlo * (f32x2){s4[g].x, s4[g].y}; ps2 += vhi * (f32x2){s4[g].z, s4[g].w};
          pd2 += vlo * (f32x2){d4[g].x, d4[g].y}; pd2 += vhi * (f32x2){d4[g].z, d4[g].w};
        }
        float ps = ps2.x + ps2.y, pd = pd2.x + pd2.y;
        asm volatile("" : "+v"(ps), "+v"(pd));      // due here, not sunk below the stores
        pts[t] = ps; ptd[t] = pd;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) *reinterpret_cast<float4 *>(patch + r * 36 + 8 * g + 4 * h) = v[g];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        *reinterpret_cast<float4 *>(dst[k] + t * 32) =
            *reinterpret_cast<const float4 *>(patch + ((lane >> 3) + 8 * k) * 36 + (lane & 7) * 4);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (ATT) {
    // per-tile partials -> heads, in ascending tile order (the order the single-pass form summed them in)
    const bool ok = row < M;
    const int H = a.H, tph = a.C / 32;              // tiles per head
    for (int hd = 0; hd < H; ++hd) {
      float ps = 0.0f, pd = 0.0f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (t / tph == hd) { ps += pts[t]; pd += ptd[t]; }
      ps += __shfl_xor(ps, 32);
      pd += __shfl_xor(pd, 32);
      if (ok && h == 0) {
        const int HT = a.asd_H ? a.asd_H : H;
        a.asd[row * 2 * HT + a.asd_hd0 + hd] = ps;
        a.asd[row * 2 * HT + HT + a.asd_hd0 + hd] = pd;
      }
    }
  }
}

// ---- W-resident persistent form (K = 64: the folded extractor-2 x lin_0 GEMM) ----------------------------------
// The whole Wt [64][NC] (64 KiB at NC = 256) is staged into LDS ONCE per workgroup; 8 waves then walk the row blocks
// (32 rows each, stride = grid) with no barrier at all: X fragments come per lane from global memory, the epilogue
// (bias / ReLU / attention dots / LDS-transposed row stores) is the one above.  Against the chunked kernel this
// drops the per-128-rows re-staging of W (65 536 x 64 KiB of L2 -> LDS traffic per launch) and every barrier.
// Ablations (BGNN_GEMM_DBG, 8.4 M rows, NC = 256): 3.5 ms as is; 2.3 ms without the MFMAs (= 11 GB at 4.8 TB/s: the
// memory side alone is HBM-bound); 1.75 ms is the MFMA floor; 4 instead of 8 waves per CU: 4.6 ms.  With two waves per
// SIMD (128 accumulators each) the two phases overlap only partly; a start stagger of the second wave changes nothing.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_lp(const bf16x8 &a, const bf16x8 &b, const f32x16 &c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_lp(const f16x8 &a, const f16x8 &b, const f32x16 &c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// SP (1 bf16x3, 2 fp16x3): operand-split matrix path (opt-in, see gat_layer_fused.hip): Wt is then the hi / lo split image of pack_split_bf16 and X
// is split in registers; lane (r, h) owns k = 16 step + 8h + i of every 16-wide k-step.
// SP = 3 (bf16 activation storage, BASELINE config 3): Wt is the hi-only bf16 image, X is rounded to bf16 in registers, one
// MFMA per tile and 16 k, and Y is written as bf16 (the attention dots are taken from the f32 accumulators).
// FRONT: the K = 64 input is not read from memory but made on the spot from the [M][8] node feature table:
//   h1 = relu(x8 @ W0^T + b0)   (feature extractor layer 1; gnn.py LocalFeatureExtractor)
// by the SAME eight v_mfma_f32_32x32x2_f32 per 32 rows (and the same k pairing, bias add and ReLU) as the stand-alone
// gemm_f32_kernel<2, false> launch it replaces -- bit-identical h1, which never goes to HBM (256 B/node written + read back,
// and one launch, less).  The result tile holds, on lane (r, h), channels 8s + 4h + i of row r: exactly this kernel's X fragment
// order on the exact path; on the bf16 path it is the accumulator-as-operand order, for which the W image is packed
// (bgnn_api.hip pack_bf16_image_accop).
// PM (exact path, attention form with the fused front, NT >= 4): tile-PAIR-major MFMA order over a weight image whose columns
// interleave the two tiles of a pair (bgnn_api.hip pack_tilegroup_image, TG = 2: one ds_read_b64 = the fragments of both tiles of a
// k row), with the epilogue of pair p -- bias, attention dots, transposed patch, row stores -- issued in eight slots BETWEEN the
// eight MFMA groups of pair p + 1.  In the tile-major form a wave's life is 256 MFMAs, then ~600 epilogue instructions; the two
// waves of a SIMD share the matrix pipe fairly, so they stay in phase -- both in their MFMAs, then both in their epilogues with
// the pipe idle (MfmaUtil 65 %).  Interleaved, a wave's instruction stream is uniform: its LDS round trips and store issue pass
// under its own MFMAs and the row stores leave at an even pace.  Every accumulator sees the same products in the same k order and
// the dots are summed in the same order: bit-identical to the tile-major form (option gemm_pair_major = 0).
template <int NT, bool ATT, int SP = 0, bool FRONT = false, bool PM = false>
__global__ __launch_bounds__(512, 2) void gemm_wres64_kernel(GemmArgs a) {
  static_assert(!FRONT || SP == 0 || SP == 3, "the fused front exists on the exact and the bf16 storage paths");
  static_assert(!PM || (SP == 0 && ATT && FRONT && NT % 2 == 0 && NT >= 4), "pair-major form: exact path, attention epilogue, fused front");
  constexpr int NC = NT * 32, K = 64;
  constexpr int WFLOATS = SP == 3 ? K * NC / 2 : K * NC;   // bytes of the weight image / 4
  extern __shared__ __attribute__((aligned(128))) float wres_lds[];
  float *wl = wres_lds;                              // [64][NC]
  constexpr int PP = 68;                              // patch pitch: two 32-column tiles side by side + 4 pad
  // bf16 output: the attention dots are a NINTH 32-column MFMA tile (AMF).  Its weight columns are W att folded on the host
  // (bgnn_api.hip pack_alpha_tile): column hd = sum over head hd's columns of W_bf16[k][c] att_src[c] (4 + hd: att_dst), as bf16
  // hi parts, columns 8.. / 12.. their bf16 lo parts, so that hi + lo carries 16 mantissa bits; the bias' share is a constant per
  // head behind the image.  Against taking the dots from the accumulators this drops 64 ds_read_b128 of the att vectors, 64
  // v_pk_fma_f32 and 8 scalar stores per 32 rows for 4 MFMAs and one 16-byte store: the launch was bound by exactly that LDS and
  // VALU work in its epilogue, not by HBM (ablations in profiles/).
  constexpr bool AMF = SP == 3 && ATT;
  constexpr int WIMG = WFLOATS + (AMF ? 1024 : 0);   // + the alpha tile: 4 k-steps x 1 KiB
  // bf16 output: the wave's patch holds FOUR tiles side by side as bf16, [32 rows][256 B]; else two as f32
  constexpr int PATCHF = SP == 3 ? 32 * 64 : 32 * PP;
  float *patches = wl + WIMG;                        // [8][PATCHF]
  float *attl = patches + 8 * PATCHF;                // [2][NC]
  float *b0l = attl + 2 * NC;                        // [64]  (FRONT)
  // The output bias lives in LDS as well: read from global memory inside the epilogue, each of its 32 float4 loads per block
  // sat -- vmcnt counts loads and stores in one in-order queue on gfx9 -- behind the row stores of the tile pair before it, so
  // the epilogue advanced at the pace of HBM write latency (~1 ms of the 3.2 ms launch).
  float *biasl = b0l + 64;                           // [NC]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int NW = blockDim.x >> 6;                    // 8 (4 only in the occupancy experiment)
  {
    const char *src = reinterpret_cast<const char *>(a.Wt);
    constexpr int NQ = WIMG * 4 / 1024;              // 1-KiB pieces
    for (int j = 0; j < (NQ + NW - 1) / NW; ++j) {
      const int q = j * NW + wave;
      if (q < NQ)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const void *>(src + q * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)(wl + q * 256), 16, 0, 0);
    }
    if (ATT && !AMF)
      for (int i = threadIdx.x; i < NC; i += blockDim.x) { attl[i] = a.att_src[i]; attl[NC + i] = a.att_dst[i]; }
    if (FRONT && threadIdx.x < 64) b0l[threadIdx.x] = a.b0[threadIdx.x];
    if (a.bias)
      for (int i = threadIdx.x; i < NC; i += blockDim.x) biasl[i] = a.bias[i];
  }
  __syncthreads();                                   // (vmcnt(0): W has landed)
  const int64_t M = *a.d_m;
  float *patch = patches + wave * PATCHF;
  float cbv[4] = {0.f, 0.f, 0.f, 0.f};               // AMF: the bias' share of this lane's four dots (h = 0: src, h = 1: dst)
  if constexpr (AMF) {
#pragma unroll
    for (int i = 0; i < 4; ++i) cbv[i] = a.Wt[WIMG + 4 * h + i];
  }
  const uint32_t asl = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)(attl + 4 * h);
  const int64_t stride = (int64_t)gridDim.x * NW * 32;
  int64_t row0 = (int64_t)blockIdx.x * NW * 32 + wave * 32;
  float4 ax[K / 8];
  f32x4 xq;                                           // FRONT: features 4h..4h+3 of the lane's row
  float w0[2][4];                                     // FRONT: W0^T[4h + i][32 t + r], resident
  if constexpr (FRONT) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) w0[t][i] = a.W0t[(4 * h + i) * 64 + t * 32 + r];
  }
  // An attention GEMM has no activation (GATConv's lin), and the fused front always carries the folded bias: known at compile time,
  // so the epilogue's 4 x NT uniform branches and NT x 16 v_max are not in the loop at all.
  const bool relu = ATT ? false : a.relu != 0;
  const bool has_bias = FRONT ? true : a.bias != nullptr;
  // FRONT: the next block's feature row is requested by an asm load the compiler does not track.  vmcnt is ONE in-order queue for
  // loads and stores on gfx9: left to hipcc, the wait for that row at the top of the loop is vmcnt(0), i.e. for every row store of
  // the block before it as well -- each wave then idles for HBM's write latency once per block.  The load is issued BEFORE the
  // epilogue's NT / 2 x 8 row-segment stores (unconditional: rows past M go to the dump row), so `vmcnt(8 * (NT / 2))` at the top of
  // the next block leaves exactly those stores in flight (the [alpha_src | alpha_dst] stores behind them, when issued, only make
  // the wait wake up earlier than strictly possible).
  constexpr int STORES_AFTER_X = 8 * (NT / 2);
  auto load_x = [&](int64_t rb) {
    const int64_t row = rb + r;
    if constexpr (FRONT) {
      const float *xp = a.X + (row < M ? row : M - 1) * 8 + 4 * h;
      if constexpr (SP == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xq) : "v"(xp) : "memory");
      else xq = *reinterpret_cast<const f32x4 *>(xp);
    } else {
      const float *xp = a.X + (row < M ? row : M - 1) * a.ldx + (SP ? 8 * h : 4 * h);
#pragma unroll
      for (int s = 0; s < K / 8; ++s)                  // exact: k = 8s + 4h + i; split: k = 16 (s/2) + 8h + 4 (s&1) + i
        ax[s] = *reinterpret_cast<const float4 *>(xp + (SP ? (s >> 1) * 16 + (s & 1) * 4 : s * 8));
    }
  };
  if (row0 < M) {
    load_x(row0);
    if constexpr (FRONT && SP == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(xq));
  }
  for (bool first = true; row0 < M; row0 += stride, first = false) {
    if constexpr (FRONT && SP == 0) {
      if (!first) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(xq) : "n"(STORES_AFTER_X));
    }
    if constexpr (FRONT) {
      // extractor layer 1 on the spot: the instruction sequence of gemm_f32_kernel<2, false> (K = 8)
      f32x16 a1[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) a1[t][i] = 0.0f;
      const float xv[4] = {xq.x, xq.y, xq.z, xq.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t) a1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[t][i], xv[i], a1[t], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < K / 8; ++s) {               // channels 8s + 4h + i = tile s / 4, registers 4 (s % 4) + i
        const float4 b = *reinterpret_cast<const float4 *>(b0l + 8 * s + 4 * h);
        float4 v = make_float4(a1[s / 4][4 * (s % 4)] + b.x, a1[s / 4][4 * (s % 4) + 1] + b.y, a1[s / 4][4 * (s % 4) + 2] + b.z,
                               a1[s / 4][4 * (s % 4) + 3] + b.w);
        v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
        ax[s] = v;
      }
    }
    float pts[ATT ? NT : 1], ptd[ATT ? NT : 1];
    if constexpr (PM) {
      constexpr int NP = NT / 2;
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      // the next block's feature row: xq is free again (the front consumed it), and every row store of this block follows
      if (row0 + stride < M && !(BGNN_DIAG && (a.dbg & 4))) load_x(row0 + stride);
      // row-segment stores: rows row0 + (lane >> 4) + 4 k -- a wave-uniform 64-bit base per k (scalar registers) plus ONE 32-bit
      // lane offset that never changes, instead of eight 64-bit pointers in vector registers.  Rows past M are masked, not sent to
      // the dump row: only the grid's very last block is partial, and it has no next block whose feature row would count on them.
      const int64_t urow0 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(row0 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)row0);
      const uint32_t ldyb = (uint32_t)a.ldy * 4u;
      const uint32_t voff = (uint32_t)(lane >> 4) * ldyb + (uint32_t)(lane & 15) * 16u;
      const int rows_left = (int)(M - urow0 < 32 ? M - urow0 : 32);     // wave-uniform
      f32x16 acc[NT];
      // pair-permuted image: column 32 t + r of row k at k NC + 64 (t / 2) + 2 r + (t & 1); lane (r, h) reads rows 8 s + 4 h + i
      const float *wlane = wl + 4 * h * NC + 2 * r;
      auto wfrag = [&](int s, int i, int p) -> f32x2 { return *reinterpret_cast<const f32x2 *>(wlane + (s * 8 + i) * NC + p * 64); };
      // epilogue of one pair in micro-stages (t: tile, hf: half of its 16 accumulator registers, kh: half of the 8 row-segment stores)
      f32x4 eb[2], es[2], ed[2], pv[4];
      f32x2 ps2 = {0.f, 0.f}, pd2 = {0.f, 0.f};
      // (the base is made opaque once per block: left visible, the loop-invariant address of every one of the 96 reads is hoisted
      //  out of the row loop into a vector register of its own -- 30 spills; inside the loop they fold into the reads' offset fields)
      typedef const __attribute__((address_space(3))) float lds_cf;
      typedef const __attribute__((address_space(3))) f32x4 lds_cf4;
      lds_cf *ebase = (lds_cf *)(attl + 4 * h);         // att_src | att_dst [2 NC] | b0 [64] | bias [NC]
      asm volatile("" : "+v"(ebase));
      auto stage_R = [&](int t, int hf) {               // bias and attention vectors of the half tile: six LDS reads in flight
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int c0 = t * 32 + 8 * (2 * hf + gg);
          eb[gg] = *reinterpret_cast<lds_cf4 *>(ebase + 2 * NC + 64 + c0);
          es[gg] = *reinterpret_cast<lds_cf4 *>(ebase + c0);
          ed[gg] = *reinterpret_cast<lds_cf4 *>(ebase + NC + c0);
        }
      };
      auto stage_C = [&](int t, int hf) {               // + bias, the dots' partial sums (same order as the tile-major form), patch
        if (hf == 0) { ps2 = (f32x2){0.f, 0.f}; pd2 = (f32x2){0.f, 0.f}; }
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int g = 2 * hf + gg;
          f32x4 v = {acc[t][4 * g] + eb[gg].x, acc[t][4 * g + 1] + eb[gg].y, acc[t][4 * g + 2] + eb[gg].z, acc[t][4 * g + 3] + eb[gg].w};
          const f32x2 vlo = {v.x, v.y}, vhi = {v.z, v.w};
          ps2 += vlo * (f32x2){es[gg].x, es[gg].y}; ps2 += vhi * (f32x2){es[gg].z, es[gg].w};
          pd2 += vlo * (f32x2){ed[gg].x, ed[gg].y}; pd2 += vhi * (f32x2){ed[gg].z, ed[gg].w};
          *reinterpret_cast<f32x4 *>(patch + r * PP + (t & 1) * 32 + 8 * g + 4 * h) = v;
        }
        if (hf == 1) {
          float ps = ps2.x + ps2.y, pd = pd2.x + pd2.y;
          asm volatile("" : "+v"(ps), "+v"(pd));
          // C == 64: a pair of tiles is a head.  Summed as the tile-major form sums them: (0 + tile 2q) + tile 2q + 1
          if ((t & 1) == 0) { pts[t / 2] = 0.0f + ps; ptd[t / 2] = 0.0f + pd; }
          else { pts[t / 2] += ps; ptd[t / 2] += pd; }
        }
      };
      auto stage_PR = [&](int kh) {                     // four 256-byte row segments of the pair's patch
#pragma unroll
        for (int j = 0; j < 4; ++j)
          pv[j] = *reinterpret_cast<const f32x4 *>(patch + ((lane >> 4) + 4 * (4 * kh + j)) * PP + (lane & 15) * 4);
      };
      auto stage_ST = [&](int q, int kh) {
        if (!(BGNN_DIAG && (a.dbg & 1))) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = 4 * kh + j;
            char *rowb = reinterpret_cast<char *>(a.Y) + (urow0 + 4 * k) * (int64_t)ldyb + q * 64 * 4;   // uniform
            if ((lane >> 4) + 4 * k < rows_left) *reinterpret_cast<f32x4 *>(rowb + voff) = pv[j];
          }
        }
      };
      auto epilogue_slot = [&](int q, int slot) {       // pair q's epilogue, slot 0 .. 7
        const int tA = 2 * q, tB = 2 * q + 1;
        if (slot == 0) { stage_R(tA, 0); }
        if (slot == 1) { stage_C(tA, 0); stage_R(tA, 1); }
        if (slot == 2) { stage_C(tA, 1); stage_R(tB, 0); }
        if (slot == 3) { stage_C(tB, 0); stage_R(tB, 1); }
        if (slot == 4) { stage_C(tB, 1); }
        if (slot == 5) { stage_PR(0); }
        if (slot == 6) { stage_ST(q, 0); stage_PR(1); }
        if (slot == 7) { stage_ST(q, 1); }
      };
      f32x2 wc[4], wn[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) wc[i] = wfrag(0, i, 0);
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int s = 0; s < K / 8; ++s) {
          if (!(p == NP - 1 && s == K / 8 - 1)) {       // the next group's fragments fly under this group's MFMAs
            const int ns = s == K / 8 - 1 ? 0 : s + 1, np = s == K / 8 - 1 ? p + 1 : p;
#pragma unroll
            for (int i = 0; i < 4; ++i) wn[i] = wfrag(ns, i, np);
          }
          const float av[4] = {ax[s].x, ax[s].y, ax[s].z, ax[s].w};
          if (!(BGNN_DIAG && (a.dbg & 2))) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (s == 0 && i == 0) {                   // the pair's first MFMAs start from an inline zero (no 32 v_mov per pair)
                acc[2 * p] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[i].x, av[i], zero16, 0, 0, 0);
                acc[2 * p + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[i].y, av[i], zero16, 0, 0, 0);
              } else {
                acc[2 * p] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[i].x, av[i], acc[2 * p], 0, 0, 0);
                acc[2 * p + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[i].y, av[i], acc[2 * p + 1], 0, 0, 0);
              }
            }
          } else if (s == 0) {
            acc[2 * p] = zero16; acc[2 * p + 1] = zero16;
          }
          __builtin_amdgcn_sched_barrier(0);
          if (p > 0) epilogue_slot(p - 1, s);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 4; ++i) wc[i] = wn[i];
        }
      }
#pragma unroll
      for (int slot = 0; slot < 8; ++slot) epilogue_slot(NP - 1, slot);   // the last pair's epilogue (the SIMD's other wave covers it)
      __builtin_amdgcn_sched_barrier(0);
    } else {
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    f32x16 acca;                                        // AMF: the alpha tile
    if constexpr (AMF) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acca[i] = 0.0f;
    }
    if constexpr (SP == 3) {
      if (!(BGNN_DIAG && (a.dbg & 2))) {
#pragma unroll
        for (int st = 0; st < K / 16; ++st) {
          const float v[8] = {ax[2 * st].x, ax[2 * st].y, ax[2 * st].z, ax[2 * st].w,
                              ax[2 * st + 1].x, ax[2 * st + 1].y, ax[2 * st + 1].z, ax[2 * st + 1].w};
          bf16x8 xh;
#pragma unroll
          for (int i = 0; i < 8; ++i) xh[i] = (__bf16)v[i];
#pragma unroll
          for (int t = 0; t < NT; ++t) {               // half-chunk st: tile t at (st * NT + t) KiB of the hi-only image
            const bf16x8 wh = *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const char *>(wl) + (st * NT + t) * 1024 + lane * 16);
            acc[t] = mfma_lp(wh, xh, acc[t]);
          }
          if constexpr (AMF) {                         // the alpha tile sits behind the K / 16 * NT KiB of W
            const bf16x8 wa = *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const char *>(wl) + (K / 16 * NT + st) * 1024 + lane * 16);
            acca = mfma_lp(wa, xh, acca);
          }
        }
      }
    } else if constexpr (SP != 0) {
      using LP8 = typename std::conditional<SP == 2, f16x8, bf16x8>::type;
      using LPE = typename std::conditional<SP == 2, _Float16, __bf16>::type;
      if (!(BGNN_DIAG && (a.dbg & 2))) {
#pragma unroll
        for (int st = 0; st < K / 16; ++st) {
          const float v[8] = {ax[2 * st].x, ax[2 * st].y, ax[2 * st].z, ax[2 * st].w,
                              ax[2 * st + 1].x, ax[2 * st + 1].y, ax[2 * st + 1].z, ax[2 * st + 1].w};
          LP8 xh, xl;
#pragma unroll
          for (int i = 0; i < 8; ++i) xh[i] = (LPE)v[i];
#pragma unroll
          for (int i = 0; i < 8; ++i) xl[i] = (LPE)(v[i] - (float)xh[i]);
#pragma unroll
          for (int t = 0; t < NT; ++t) {               // half-chunk st: tile t, part p at (st * NT * 2 + 2t + p) KiB
            const LP8 wh = *reinterpret_cast<const LP8 *>(reinterpret_cast<const char *>(wl) + ((st * NT + t) * 2) * 1024 + lane * 16);
            const LP8 wlo = *reinterpret_cast<const LP8 *>(reinterpret_cast<const char *>(wl) + ((st * NT + t) * 2 + 1) * 1024 + lane * 16);
            acc[t] = mfma_lp(wlo, xh, acc[t]);
            acc[t] = mfma_lp(wh, xl, acc[t]);
            acc[t] = mfma_lp(wh, xh, acc[t]);
          }
        }
        if constexpr (SP == 2) {                       // the image holds W * 2^S (bgnn_api.hip pack_split): exact power-of-two rescale
          const float wi = a.w_inv;
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] *= wi;
        }
      }
    } else if (!(BGNN_DIAG && (a.dbg & 2)))
#pragma unroll
    for (int s = 0; s < K / 8; ++s) {
      const float av[4] = {ax[s].x, ax[s].y, ax[s].z, ax[s].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float *wrow = wl + (s * 8 + 4 * h + i) * NC + r;
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[t * 32], av[i], acc[t], 0, 0, 0);
      }
    }
    if (row0 + stride < M && !(BGNN_DIAG && (a.dbg & 4))) load_x(row0 + stride);    // next block's X flies under this block's epilogue
    // stores leave as 256-byte row segments (two tiles side by side in the patch): 16 lanes x 16 B per row, 4 rows per
    // instruction -- half as many separate DRAM bursts per 1-KiB output row as 128-byte segments
    static_assert(NT % 2 == 0, "tiles are stored in pairs (bf16: in fours, the last group may be short)");
    constexpr int YB = SP == 3 ? 2 : 4;              // bytes per stored output element
    // (bf16: FOUR tiles side by side in the patch, already as bf16 -- half the LDS write bytes, half the read-backs, and the same
    //  256-byte row segments leave as 16 bytes per lane instead of 8: half the store instructions)
    constexpr int LB = SP == 3 ? 16 : 4 * YB;         // bytes per lane of a row-segment store
    char *dst[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t rr = row0 + (lane >> 4) + 4 * k;
      dst[k] = (rr < M ? reinterpret_cast<char *>(a.Y) + rr * a.ldy * YB : reinterpret_cast<char *>(a.dump)) + (lane & 15) * LB;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float4 v[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        v[g] = make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
        if (has_bias) {
          const float4 b = *reinterpret_cast<const float4 *>(biasl + t * 32 + 8 * g + 4 * h);
          v[g].x += b.x; v[g].y += b.y; v[g].z += b.z; v[g].w += b.w;
        }
        if (relu) {
          v[g].x = v[g].x > 0.f ? v[g].x : 0.f; v[g].y = v[g].y > 0.f ? v[g].y : 0.f;
          v[g].z = v[g].z > 0.f ? v[g].z : 0.f; v[g].w = v[g].w > 0.f ? v[g].w : 0.f;
        }
      }
      if (ATT && !AMF) {
        f32x4 s4[4], d4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          asm volatile("ds_read_b128 %0, %1" : "=v"(s4[g]) : "v"(asl + (t * 32 + 8 * g) * 4));
          asm volatile("ds_read_b128 %0, %1" : "=v"(d4[g]) : "v"(asl + (NC + t * 32 + 8 * g) * 4));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 ps2 = {0.f, 0.f}, pd2 = {0.f, 0.f};       // two-lane partial sums: v_pk_fma_f32
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x2 vlo = {v[g].x, v[g].y}, vhi = {v[g].z, v[g].w};
          ps2 += vlo * (f32x2){s4[g].x, s4[g].y}; ps2 += vhi * (f32x2){s4[g].z, s4[g].w};
          pd2 += vlo * (f32x2){d4[g].x, d4[g].y}; pd2 += vhi * (f32x2){d4[g].z, d4[g].w};
        }
        float ps = ps2.x + ps2.y, pd = pd2.x + pd2.y;
        asm volatile("" : "+v"(ps), "+v"(pd));
        pts[t] = ps; ptd[t] = pd;
      }
      if constexpr (SP == 3) {
        // Patch layout [32 rows][256 B]: 16-byte chunk c (8 columns) of row r lives at chunk c ^ (r & 15), and the two 8-byte halves
        // of a chunk are swapped on rows with bit 3 set.  ds_write_b64 is served in groups of 16 consecutive lanes over 32 banks:
        // the 16 rows of a group then hit 16 distinct bank pairs (chunk' mod 8 twice, with different halves); the ds_read_b128 below
        // is served in 16-lane groups that pair chunks {0-3, 12-15} of one row with {4-11} of the next (MI355X_MICROARCH.md): the
        // XOR only permutes inside those sets (the rows of one instruction share the bits above 1), so the 256-byte pitch reads
        // without conflicts too.
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        char *pb = reinterpret_cast<char *>(patch) + r * 256 + ((h ^ ((r >> 3) & 1)) << 3);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 o;
          o[0] = (__bf16)v[g].x; o[1] = (__bf16)v[g].y; o[2] = (__bf16)v[g].z; o[3] = (__bf16)v[g].w;
          *reinterpret_cast<bf16x4 *>(pb + ((((t & 3) * 4 + g) ^ (r & 15)) << 4)) = o;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (((t & 3) == 3 || t == NT - 1) && !(BGNN_DIAG && (a.dbg & 1))) {
          constexpr int NCH = ((NT - 1) & 3) * 4 + 4;  // chunks per row in the LAST flush (NT = 2: 8 of the 16 lanes of a row store)
          const bool full = (t & 3) == 3;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int row = (lane >> 4) + 4 * k;       // (row >> 3) & 1 == (k >> 1) & 1: the half swap is known at compile time
            uint4 q = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(patch) + row * 256 + (((lane & 15) ^ (row & 15)) << 4));
            if ((k >> 1) & 1) q = make_uint4(q.z, q.w, q.x, q.y);
            if (full || (lane & 15) < NCH) *reinterpret_cast<uint4 *>(dst[k] + (t & ~3) * 32 * YB) = q;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<float4 *>(patch + r * PP + (t & 1) * 32 + 8 * g + 4 * h) = v[g];
        __builtin_amdgcn_sched_barrier(0);
        if ((t & 1) && !(BGNN_DIAG && (a.dbg & 1))) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float4 v4 = *reinterpret_cast<const float4 *>(patch + ((lane >> 4) + 4 * k) * PP + (lane & 15) * 4);
            *reinterpret_cast<float4 *>(dst[k] + (t - 1) * 32 * YB) = v4;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if constexpr (AMF) {
      // lane (r, h): registers 0-3 = the hi parts of head 0-3's src (h = 0) / dst (h = 1) dot, registers 4-7 the lo parts
      const int64_t row = row0 + r;
      const int H = a.H;
      if (row < M) {
        const float d0 = acca[0] + acca[4] + cbv[0], d1 = acca[1] + acca[5] + cbv[1], d2 = acca[2] + acca[6] + cbv[2],
                    d3 = acca[3] + acca[7] + cbv[3];
        float *p = a.asd + row * 2 * H + h * H;
        if (H == 4) *reinterpret_cast<float4 *>(p) = make_float4(d0, d1, d2, d3);
        else { p[0] = d0; if (H > 1) p[1] = d1; if (H > 2) p[2] = d2; }
      }
    }
    }   // !PM
    if constexpr (PM) {
      const int64_t row = row0 + r;
#pragma unroll
      for (int hd = 0; hd < NT / 2; ++hd) {
        const float ps = pts[hd] + __shfl_xor(pts[hd], 32), pd = ptd[hd] + __shfl_xor(ptd[hd], 32);
        if (row < M && h == 0) {
          a.asd[row * NT + hd] = ps;                   // [alpha_src (H) | alpha_dst (H)], H = NT / 2
          a.asd[row * NT + NT / 2 + hd] = pd;
        }
      }
    } else if (ATT && !AMF) {
      const int64_t row = row0 + r;
      const int H = a.H, tph = a.C / 32;
      for (int hd = 0; hd < H; ++hd) {
        float ps = 0.0f, pd = 0.0f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (t / tph == hd) { ps += pts[t]; pd += ptd[t]; }
        ps += __shfl_xor(ps, 32);
        pd += __shfl_xor(pd, 32);
        if (row < M && h == 0) {
          a.asd[row * 2 * H + hd] = ps;
          a.asd[row * 2 * H + H + hd] = pd;
        }
      }
    }
  }
}

template <int NT, bool ATT, int SP = 0, bool FRONT = false, bool PM = false>
static int launch_wres64(bgnn_ctx *ctx, const GemmArgs &a) {
  constexpr size_t lds_bytes = (size_t)((SP == 3 ? 32 : 64) * NT * 32 + (SP == 3 && ATT ? 1024 : 0) + 8 * (SP == 3 ? 32 * 64 : 32 * 68) +
                                        2 * NT * 32 + 64 + NT * 32) * 4;
  static std::atomic<uint64_t> configured{0};   // per instantiation: one bit per device (the attribute is per device)
  auto kern = gemm_wres64_kernel<NT, ATT, SP, FRONT, PM>;
  if (!(configured.load(std::memory_order_relaxed) >> (ctx->device & 63) & 1)) {
    BGNN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    configured.fetch_or(1ull << (ctx->device & 63), std::memory_order_relaxed);
  }
  const int per_cu = lds_bytes > 80 * 1024 ? 1 : 2;
  const int nw = ctx->opts.gemm_waves;
  hipLaunchKernelGGL(kern, dim3(ctx->num_cus * per_cu), dim3(64 * nw), lds_bytes, ctx->stream, a);
  BGNN_HIP_CHECK(hipGetLastError());
  return BGNN_OK;
}

// ---- layer 0 "aggregate first" (matrix_path = bf16, gat_layer_fused.hip gat_layer_bf16_2p_kernel<K, 4, AF>) --------------------------
// GATConv's sum over the in-edges is linear, so  sum_j alpha_ij (W h1_j + b)  =  W (sum_j alpha_ij h1_j) + b sum_j alpha_ij : layer 0 can
// aggregate the extractor's 64-channel h1 (128 bytes per node as bf16) instead of the 256-channel lin_0 product (512 bytes) and apply
// each head's 64 x 64 block of the folded lin_0 weight afterwards, inside the fused launch.  What is left of the front GEMM is this
// kernel: extractor layer 1 (the same eight float32 MFMAs), h1 rounded to bf16 -- the operand the front GEMM multiplies -- stored, and the
// attention dots of layer 0 through the front GEMM's own alpha tile (pack_alpha_tile): [alpha_src | alpha_dst] come out bit for bit
// as the bf16 front GEMM writes them.  32 + 128 + 32 bytes per node instead of 32 + 512 + 32, no 64 -> 256 product.
struct ExtractorAfArgs {
  const float *X;         // [M][8] node features
  const float *W0t;       // [8][64] extractor layer 1 (transposed)
  const float *b0;        // [64]
  const float *alpha_tile;   // 4 k-steps x 1 KiB of bf16 A fragments, then 8 floats: the bias' share of the dots
  void *h1;               // [M][64] bf16
  float *asd;             // [M][2 H]
  const int64_t *d_m;
  int H;
};

__global__ __launch_bounds__(256, 3) void extractor_af_kernel(ExtractorAfArgs a) {
  __shared__ __attribute__((aligned(128))) char patches_[4 * 32 * 128];     // per wave: [32 rows][128 B] of bf16, chunks XOR-swizzled
  __shared__ float b0l[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  if (threadIdx.x < 64) b0l[threadIdx.x] = a.b0[threadIdx.x];
  float w0[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) w0[t][i] = a.W0t[(4 * h + i) * 64 + t * 32 + r];
  bf16x8 wa[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) wa[st] = *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const char *>(a.alpha_tile) + st * 1024 + lane * 16);
  float cbv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) cbv[i] = a.alpha_tile[1024 + 4 * h + i];
  __syncthreads();
  const int64_t M = *a.d_m;
  char *patch = patches_ + wave * (32 * 128);
  const int64_t stride = (int64_t)gridDim.x * 4 * 32;
  int64_t row0 = (int64_t)blockIdx.x * 4 * 32 + wave * 32;
  f32x4 xq = {0.f, 0.f, 0.f, 0.f};
  auto load_x = [&](int64_t rb) {
    const int64_t row = rb + r;
    xq = *reinterpret_cast<const f32x4 *>(a.X + (row < M ? row : M - 1) * 8 + 4 * h);
  };
  if (row0 < M) load_x(row0);
  for (; row0 < M; row0 += stride) {
    // extractor layer 1: the instruction sequence of the front GEMM (gemm_wres64_kernel, FRONT)
    f32x16 a1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) a1[t][i] = 0.0f;
    const float xv[4] = {xq.x, xq.y, xq.z, xq.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < 2; ++t) a1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[t][i], xv[i], a1[t], 0, 0, 0);
    if (row0 + stride < M) load_x(row0 + stride);
    float4 ax[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {                     // channels 8s + 4h + i = tile s / 4, registers 4 (s % 4) + i
      const float4 b = *reinterpret_cast<const float4 *>(b0l + 8 * s + 4 * h);
      float4 v = make_float4(a1[s / 4][4 * (s % 4)] + b.x, a1[s / 4][4 * (s % 4) + 1] + b.y, a1[s / 4][4 * (s % 4) + 2] + b.z,
                             a1[s / 4][4 * (s % 4) + 3] + b.w);
      v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
      ax[s] = v;
    }
    f32x16 acca;
#pragma unroll
    for (int i = 0; i < 16; ++i) acca[i] = 0.0f;
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    // patch: 16-byte chunk s (channels 8s .. 8s + 7) of row r at chunk s ^ (r & 7), its two 8-byte halves swapped on rows with bit 3 set
    // (ds_write_b64 is served per 16 consecutive lanes: 8 chunk slots x 2 halves = 16 distinct bank pairs)
    char *pb = patch + r * 128 + ((h ^ ((r >> 3) & 1)) << 3);
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const float v[8] = {ax[2 * st].x, ax[2 * st].y, ax[2 * st].z, ax[2 * st].w, ax[2 * st + 1].x, ax[2 * st + 1].y, ax[2 * st + 1].z, ax[2 * st + 1].w};
      bf16x8 xh;
#pragma unroll
      for (int i = 0; i < 8; ++i) xh[i] = (__bf16)v[i];
      acca = mfma_lp(wa[st], xh, acca);
      const bf16x4 lo = {xh[0], xh[1], xh[2], xh[3]}, hi = {xh[4], xh[5], xh[6], xh[7]};
      *reinterpret_cast<bf16x4 *>(pb + (((2 * st) ^ (r & 7)) << 4)) = lo;
      *reinterpret_cast<bf16x4 *>(pb + (((2 * st + 1) ^ (r & 7)) << 4)) = hi;
    }
    {
      // lane (r, h): registers 0-3 = the hi parts of head 0-3's src (h = 0) / dst (h = 1) dot, registers 4-7 the lo parts
      const int64_t row = row0 + r;
      const int H = a.H;
      if (row < M) {
        const float d0 = acca[0] + acca[4] + cbv[0], d1 = acca[1] + acca[5] + cbv[1], d2 = acca[2] + acca[6] + cbv[2],
                    d3 = acca[3] + acca[7] + cbv[3];
        float *p = a.asd + row * 2 * H + h * H;
        if (H == 4) *reinterpret_cast<float4 *>(p) = make_float4(d0, d1, d2, d3);
        else { p[0] = d0; if (H > 1) p[1] = d1; if (H > 2) p[2] = d2; }
      }
    }
    // (wave-private patch: the wave's own LDS writes are complete before its reads are served -- one queue, in order)
    char *dst = reinterpret_cast<char *>(a.h1) + row0 * 128 + lane * 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int row = (lane >> 3) + 8 * k;             // (row >> 3) & 1 == k & 1: the half swap is known at compile time
      uint4 q = *reinterpret_cast<const uint4 *>(patch + row * 128 + (((lane & 7) ^ (row & 7)) << 4));
      if (k & 1) q = make_uint4(q.z, q.w, q.x, q.y);
      if (row0 + row < M) *reinterpret_cast<uint4 *>(dst + k * 1024) = q;
    }
  }
}

int launch_extractor_af(bgnn_ctx *ctx, const float *x8, const float *W0t, const float *b0, const float *alpha_tile, void *h1, float *asd,
                        const int64_t *d_m, int64_t max_rows, int H) {
  if (max_rows <= 0) return BGNN_OK;
  ProfScope ps(ctx, BGNN_K_GEMM);
  ExtractorAfArgs a{x8, W0t, b0, alpha_tile, h1, asd, d_m, H};
  const int64_t groups = (max_rows + 127) / 128;
  const int grid = (int)std::min<int64_t>(groups, (int64_t)ctx->num_cus * 8);
  hipLaunchKernelGGL(extractor_af_kernel, dim3(grid), dim3(256), 0, ctx->stream, a);
  BGNN_HIP_CHECK(hipGetLastError());
  return BGNN_OK;
}

// (split_mode 3: Y receives bf16 [M][ldy], see gemm_wres64_kernel)
// Can the K = 64 attention GEMM of this shape take the feature extractor's first Linear in front (X = the [M][8] feature table)?
// From how many rows the exact path's K = 64 attention GEMM takes its W-resident form (and with it the extractor's first layer in
// front: one launch instead of two).  32 768 so that a 50 000-node batch of refinement grids is covered (65 536 until round 3:
// configs[3] 153 -> 157 M nodes/s on one context, 195 -> 203 M with two batches in flight); the results are bit-identical either way.
#ifndef BGNN_WRES_MIN_ROWS
#define BGNN_WRES_MIN_ROWS 32768
#endif
bool gemm_front_available(const bgnn_ctx *ctx, int64_t max_rows, int NC, int split_mode) {
  if (!ctx->opts.fused_front || (NC != 64 && NC != 256)) return false;
  if (split_mode == 3) return true;                                  // bf16 output: always the W-resident form
  return split_mode == 0 && !ctx->opts.gemm_no_wres && max_rows >= BGNN_WRES_MIN_ROWS;
}

int launch_gemm_f32(bgnn_ctx *ctx, const float *X, int ldx, const float *Wt, const float *bias, float *Y, int ldy,
                    const int64_t *d_m, int64_t max_rows, int K, int NC, int relu, const float *att_src,
                    const float *att_dst, float *asd, int H, int C, const float *Wt_split, int split_mode,
                    const float *front_W0t, const float *front_b0, const float *Wt_pm, const float *Wt_blk, float split_inv_scale) {
  if (NC > 256) {
    // a layer wider than 256 columns (heads x hidden up to 512): one launch of the generic kernel per 256-column block of the
    // blocked weight image; a block's heads write their attention dots into their columns of the shared [M][2 H] table.  The opt-in
    // operand-split paths (split_mode 1 / 2) have no instance here and run exact f32, like every other shape without one.
    BGNN_REQUIRE(split_mode != 3, "gemm_f32: NC=%d has no bf16-output form", NC);
    BGNN_REQUIRE(Wt_blk && NC % 256 == 0 && NC <= 512 && !front_W0t && (!att_src || (C > 0 && 256 % C == 0)),
                 "gemm_f32: NC=%d needs the blocked weight image (no fused front)", NC);
    for (int b = 0; b < NC / 256; ++b) {
      ProfScope ps(ctx, BGNN_K_GEMM);
      GemmArgs a{X, Wt_blk + (size_t)b * K * 256, bias ? bias + b * 256 : nullptr, Y + b * 256, d_m, att_src ? att_src + b * 256 : nullptr,
                 att_dst ? att_dst + b * 256 : nullptr, asd, ctx->zero_page + 2048, ldx, ldy, K, relu, att_src ? 256 / C : 0, C, 0, nullptr, nullptr,
                 att_src ? H : 0, att_src ? b * (256 / C) : 0, 1.0f};
      if (max_rows <= 0) return BGNN_OK;
      dim3 grid((unsigned)((max_rows + 127) / 128)), block(256);
      if (att_src) hipLaunchKernelGGL((gemm_f32_kernel<8, true>), grid, block, 0, ctx->stream, a);
      else hipLaunchKernelGGL((gemm_f32_kernel<8, false>), grid, block, 0, ctx->stream, a);
      BGNN_HIP_CHECK(hipGetLastError());
    }
    return BGNN_OK;
  }
  BGNN_REQUIRE(K % 8 == 0 && NC % 32 == 0 && NC <= 256 && ldx % 4 == 0 && ldy % 4 == 0,
               "gemm_f32: unsupported shape K=%d NC=%d ldx=%d ldy=%d", K, NC, ldx, ldy);
  if (att_src) BGNN_REQUIRE(C % 32 == 0 && H * C == NC, "gemm_f32: attention epilogue needs NC == H*C, C %% 32 == 0");
  if (max_rows <= 0) return BGNN_OK;
  ProfScope ps(ctx, BGNN_K_GEMM);
  const int gemm_dbg = BGNN_DIAG ? ctx->opts.gemm_diag : 0;
  GemmArgs a{X, Wt, bias, Y, d_m, att_src, att_dst, asd, ctx->zero_page + 2048, ldx, ldy, K, relu, H, C, gemm_dbg, front_W0t, front_b0, 0, 0,
             split_mode == 2 ? split_inv_scale : 1.0f};
  if (front_W0t) BGNN_REQUIRE(K == 64 && att_src && gemm_front_available(ctx, max_rows, NC, split_mode) && (split_mode != 3 || Wt_split),
                              "gemm: the fused front needs the W-resident K = 64 attention form");
  const bool no_wres = ctx->opts.gemm_no_wres != 0;
  // the split image is only read by the W-resident ATT form (NC 64 or 256); with a split path switched on that form runs
  // at EVERY batch size, so that a node's result does not depend on how many other nodes share its batch
  if (split_mode == 3) BGNN_REQUIRE(K == 64 && att_src && (NC == 64 || NC == 256) && Wt_split, "gemm: bf16 output needs the W-resident form");
  if (!(K == 64 && (!no_wres || split_mode == 3) && att_src && (NC == 64 || NC == 256))) Wt_split = nullptr;
  if (K == 64 && (!no_wres || split_mode == 3) && (max_rows >= BGNN_WRES_MIN_ROWS || Wt_split)) {    // W-resident persistent form
    switch (NC / 32) {
#define BGNN_WRES_CASE(NT) case NT:                                                                                 \
        if (front_W0t) {                                                                                           \
          if (split_mode == 3) { a.Wt = Wt_split; return launch_wres64<NT, true, 3, true>(ctx, a); }              \
          if constexpr (NT >= 4) { if (Wt_pm && ctx->opts.gemm_pair_major && C == 64 && H == NT / 2) { a.Wt = Wt_pm; return launch_wres64<NT, true, 0, true, true>(ctx, a); } } \
          return launch_wres64<NT, true, 0, true>(ctx, a);                                                        \
        }                                                                                                          \
        if (Wt_split) { BGNN_REQUIRE(split_mode != 3, "gemm: the bf16 image is packed for the fused front"); a.Wt = Wt_split; return split_mode == 2 ? launch_wres64<NT, true, 2>(ctx, a) : launch_wres64<NT, true, 1>(ctx, a); } \
        return att_src ? launch_wres64<NT, true>(ctx, a) : launch_wres64<NT, false>(ctx, a);
      BGNN_WRES_CASE(2) BGNN_WRES_CASE(8)
#undef BGNN_WRES_CASE
      default: break;
    }
  }
  dim3 grid((unsigned)((max_rows + 127) / 128)), block(256);
#define BGNN_GEMM_CASE(NT)                                                                         \
  case NT:                                                                                         \
    if (att_src) hipLaunchKernelGGL((gemm_f32_kernel<NT, true>), grid, block, 0, ctx->stream, a);  \
    else hipLaunchKernelGGL((gemm_f32_kernel<NT, false>), grid, block, 0, ctx->stream, a);         \
    break;
  switch (NC / 32) {
    BGNN_GEMM_CASE(1) BGNN_GEMM_CASE(2) BGNN_GEMM_CASE(3) BGNN_GEMM_CASE(4) BGNN_GEMM_CASE(5) BGNN_GEMM_CASE(6)
    BGNN_GEMM_CASE(7) BGNN_GEMM_CASE(8)
  }
#undef BGNN_GEMM_CASE
  BGNN_HIP_CHECK(hipGetLastError());
  return BGNN_OK;
}

}  // namespace bgnn
