// Training-mode dropout on an activation table (bgnn.h: bgnn_forward_train_dropout): the reference's nn.Dropout / F.dropout
// (models/gnn.py:57, :186, :206, :229, :253) with a counter-based Bernoulli draw per element instead of torch's generator stream.
// HBM-bound elementwise pass, 8 B per element; one 64-bit mix per element.
#include "bgnn_internal.h"

namespace bgnn {

__global__ __launch_bounds__(256) void dropout_kernel(float *x, int width, int ld, const int64_t *d_m, DropSpec d) {
  const int64_t n = *d_m * (int64_t)width;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = e / width;
    const int col = (int)(e - row * width);
    float *p = x + row * ld + col;
    *p = bgnn_drop_hash(d.seed, d.stream, (uint64_t)e) >= d.thr ? *p * d.scale : 0.0f;
  }
}

int launch_dropout(bgnn_ctx *ctx, float *x, int width, int ld, const int64_t *d_m, int64_t max_rows, const DropSpec &d) {
  if (d.thr == 0 || max_rows <= 0) return BGNN_OK;
  const int64_t n = max_rows * (int64_t)width;
  const int64_t blocks = std::min<int64_t>((n + 255) / 256, (int64_t)ctx->num_cus * 16);
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, x, width, ld, d_m, d);
  BGNN_HIP_CHECK(hipGetLastError());
  return BGNN_OK;
}

}  // namespace bgnn
