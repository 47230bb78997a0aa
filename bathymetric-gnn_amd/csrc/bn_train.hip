// BatchNorm with the statistics of the batch (the reference's backbone in train() mode, models/gnn.py:151-154
// and :179-186 -- torch_geometric BatchNorm wraps torch.nn.BatchNorm1d(eps, momentum=0.1); with the module in training
// mode it normalises with the batch mean and the BIASED batch variance and moves the running statistics towards the
// batch mean / UNBIASED variance).  z [M][W] is the convolution output (bias included) of every node of the batch:
//   pass 1  column sums and sums of squares in float64, per block of rows, no atomics  -> partial[NB][W][2]
//   pass 2  one workgroup adds the partials in block order (deterministic), derives scale / shift and the statistics
//   pass 3  z = relu?(z * scale + shift) in place
// HBM bound: z is read twice and written once (3 * 4 * W bytes per node and layer).
#include "bgnn_internal.h"

namespace bgnn {

constexpr int BN_MAX_BLOCKS = 1024;

__global__ __launch_bounds__(256) void bn_column_partial_kernel(const float *z, int ld, int W, const int64_t *d_m,
                                                                double *partial) {
  const int64_t M = *d_m;
  const int lanes = 256 / W;                       // row lanes per block (W is 32, 64, 128 or 256)
  const int col = threadIdx.x % W, rl = threadIdx.x / W;
  const int64_t per_block = (M + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * per_block, r1 = r0 + per_block < M ? r0 + per_block : M;
  double s = 0.0, q = 0.0;
  for (int64_t r = r0 + rl; r < r1; r += lanes) {
    const double v = (double)z[r * ld + col];
    s += v; q += v * v;
  }
  __shared__ double sh[2][256];
  sh[0][threadIdx.x] = s; sh[1][threadIdx.x] = q;
  __syncthreads();
  if (rl == 0) {
    for (int k = 1; k < lanes; ++k) { s += sh[0][k * W + col]; q += sh[1][k * W + col]; }
    partial[((int64_t)blockIdx.x * W + col) * 2 + 0] = s;
    partial[((int64_t)blockIdx.x * W + col) * 2 + 1] = q;
  }
}

__global__ __launch_bounds__(256) void bn_coefficients_kernel(const double *partial, int nb, int W, const int64_t *d_m,
                                                              const float *bn_w, const float *bn_b, double eps,
                                                              float *scale, float *shift, float *batch_mean,
                                                              float *batch_var_unbiased) {
  const int c = threadIdx.x;
  if (c >= W) return;
  const double M = (double)*d_m;
  double s = 0.0, q = 0.0;
  for (int b = 0; b < nb; ++b) { s += partial[((int64_t)b * W + c) * 2]; q += partial[((int64_t)b * W + c) * 2 + 1]; }
  const double mean = s / M;
  double var = q / M - mean * mean;                // biased; float64 sums of float32 data: no cancellation to speak of
  if (var < 0.0) var = 0.0;
  const double sc = (double)bn_w[c] / sqrt(var + eps);
  scale[c] = (float)sc;
  shift[c] = (float)((double)bn_b[c] - mean * sc);
  if (batch_mean) batch_mean[c] = (float)mean;
  if (batch_var_unbiased) batch_var_unbiased[c] = (float)(var * (M / (M - 1.0)));
}

__global__ __launch_bounds__(256) void bn_affine_kernel(float *z, int ld, int W, const int64_t *d_m, const float *scale,
                                                        const float *shift, int relu) {
  const int64_t M = *d_m;
  const int w4 = W / 4;
  const int64_t n = M * w4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t r = i / w4;
    const int c = (int)(i % w4) * 4;
    float4 v = *reinterpret_cast<float4 *>(z + r * ld + c);
    const float4 sc = *reinterpret_cast<const float4 *>(scale + c);
    const float4 sh = *reinterpret_cast<const float4 *>(shift + c);
    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
    if (relu) {
      v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
    }
    *reinterpret_cast<float4 *>(z + r * ld + c) = v;
  }
}

// workspace: BN_MAX_BLOCKS * W * 2 doubles of partials, then scale[W] and shift[W]
size_t bn_train_workspace_bytes(int W) { return (size_t)BN_MAX_BLOCKS * W * 2 * sizeof(double) + 2 * (size_t)W * sizeof(float); }

int launch_bn_train(bgnn_ctx *ctx, float *z, int ld, int W, int64_t max_rows, const int64_t *d_m, const float *bn_w,
                    const float *bn_b, float eps, int relu, void *workspace, float *batch_mean, float *batch_var_unbiased) {
  if (max_rows <= 0) return BGNN_OK;
  BGNN_REQUIRE(W == 32 || W == 64 || W == 128 || W == 256, "batch-statistics BatchNorm: width %d unsupported", W);
  double *partial = (double *)workspace;
  float *scale = (float *)(partial + (size_t)BN_MAX_BLOCKS * W * 2), *shift = scale + W;
  int nb = (int)((max_rows + 255) / 256);
  if (nb > BN_MAX_BLOCKS) nb = BN_MAX_BLOCKS;
  ProfScope ps(ctx, BGNN_K_AGGREGATE);
  hipLaunchKernelGGL(bn_column_partial_kernel, dim3(nb), dim3(256), 0, ctx->stream, z, ld, W, d_m, partial);
  hipLaunchKernelGGL(bn_coefficients_kernel, dim3(1), dim3(256), 0, ctx->stream, partial, nb, W, d_m, bn_w, bn_b, (double)eps,
                     scale, shift, batch_mean, batch_var_unbiased);
  int64_t blocks = (max_rows * (W / 4) + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(bn_affine_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, z, ld, W, d_m, scale, shift, relu);
  BGNN_HIP_CHECK(hipGetLastError());
  return BGNN_OK;
}

}  // namespace bgnn
