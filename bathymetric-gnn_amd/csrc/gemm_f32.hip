// K3: dense node-feature x weight GEMM in exact float32 on the matrix cores (gfx950).
//
//   Y[M, NC] = act( X[M, K] @ Wt[K, NC] + bias )
//
// Replaces the `lin(x)` GEMM of torch_geometric GATConv (reference models/gnn.py:176) and the
// Linear layers of the feature extractor / heads (models/gnn.py:52-68, 203-208).
// v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf chain, so results stay within float32
// rounding of the reference's CPU sgemm (no bf16 / xf32 shortcut exists or is wanted here).
//
// Decomposition: one wave owns 32 rows x all NC columns (NT = NC/32 accumulator tiles);
// a 256-thread workgroup = 4 waves = 128 rows.  The A operand needs one f32 per lane
// (lane l: row l&31, k-slot l>>5): each lane reads 16 contiguous bytes of ITS OWN row straight
// from global memory and the two lane halves are assigned k = 4h..4h+3 of every 8-wide k-step,
// so no LDS transpose of X is needed.  The shared Wt chunk (32 k x NC) is staged through LDS and
// read conflict-free (lanes 0-31 read 32 consecutive floats).
#include "bgnn_internal.h"

namespace bgnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GEMM_KC = 32;   // k-chunk staged in LDS

template <int NT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float *__restrict__ X, int ldx,
                                                       const float *__restrict__ Wt,
                                                       const float *__restrict__ bias, float *__restrict__ Y,
                                                       int ldy, const int64_t *__restrict__ d_m, int K, int relu) {
  constexpr int NC = NT * 32;
  __shared__ float wl[GEMM_KC * NC];
  const int64_t M = *d_m;
  const int64_t row_block = (int64_t)blockIdx.x * 128;
  if (row_block >= M) return;                       // uniform per block
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  int64_t row = row_block + wave * 32 + r;
  const int64_t row_ld = row < M ? row : M - 1;     // clamp loads, mask stores
  const float *xp = X + row_ld * ldx + 4 * h;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

  for (int kc = 0; kc < K; kc += GEMM_KC) {
    const int kn = (K - kc) < GEMM_KC ? (K - kc) : GEMM_KC;   // multiple of 8
    // A fragments for this chunk: issue before staging W so the latency overlaps
    float4 a[GEMM_KC / 8];
#pragma unroll
    for (int s = 0; s < GEMM_KC / 8; ++s)
      if (s * 8 < kn) a[s] = *reinterpret_cast<const float4 *>(xp + kc + s * 8);
    __syncthreads();                                // previous chunk fully consumed
    {
      const float4 *src = reinterpret_cast<const float4 *>(Wt + (int64_t)kc * NC);
      float4 *dst = reinterpret_cast<float4 *>(wl);
      const int n4 = kn * NC / 4;
      for (int i = threadIdx.x; i < n4; i += 256) dst[i] = src[i];
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < GEMM_KC / 8; ++s) {
      if (s * 8 < kn) {
        const float av[4] = {a[s].x, a[s].y, a[s].z, a[s].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float *wrow = wl + (s * 8 + 4 * h + i) * NC + r;
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], wrow[t * 32], acc[t], 0, 0, 0);
        }
      }
    }
  }
  // epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int64_t wrow0 = row_block + wave * 32;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = t * 32 + r;
    const float b = bias ? bias[col] : 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t orow = wrow0 + (i & 3) + 8 * (i >> 2) + 4 * h;
      float v = acc[t][i] + b;
      if (relu) v = v > 0.0f ? v : 0.0f;
      if (orow < M) Y[orow * ldy + col] = v;
    }
  }
}

int launch_gemm_f32(bgnn_ctx *ctx, const float *X, int ldx, const float *Wt, const float *bias, float *Y, int ldy,
                    const int64_t *d_m, int64_t max_rows, int K, int NC, int relu) {
  BGNN_REQUIRE(K % 8 == 0 && NC % 32 == 0 && NC <= 256 && ldx % 4 == 0, "gemm_f32: unsupported shape K=%d NC=%d ldx=%d",
               K, NC, ldx);
  if (max_rows <= 0) return BGNN_OK;
  ProfScope ps(ctx, BGNN_K_GEMM);
  dim3 grid((unsigned)((max_rows + 127) / 128)), block(256);
#define BGNN_GEMM_CASE(NT)                                                                                   \
  case NT:                                                                                                   \
    hipLaunchKernelGGL(gemm_f32_kernel<NT>, grid, block, 0, ctx->stream, X, ldx, Wt, bias, Y, ldy, d_m, K, relu); \
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
