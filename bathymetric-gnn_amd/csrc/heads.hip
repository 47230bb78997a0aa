// K5: output heads, second half.  The three first-layer Linears (64->32, ReLU) of the
// classification / confidence / correction heads (reference models/gnn.py:203-208, 226-232,
// 250-256) run as ONE GEMM into `hid` [N, 3*32 (padded to 32s)]; this kernel finishes each
// head (32->classes, 32->1 sigmoid, 32->1), then softmax / argmax (gnn.py:392-395) and the
// deployment flags of BathymetricGNN.predict (gnn.py:436-449).  One thread per node.
#include "bgnn_internal.h"

namespace bgnn {

struct HeadArgs {
  const float *hid;   // [N][ldh]
  int ldh;
  const float *W1;    // cls [classes][hh], conf [hh], corr [hh]
  const float *b1;    // cls [classes], conf [1], corr [1]
  const int64_t *d_m;
  int hh;             // hidden/2
  int classes;
  int has_corr;
  float thr_auto, thr_review;
  bgnn_outputs o;
};

constexpr int MAX_CLASSES = 16;

__global__ __launch_bounds__(256) void heads_final_kernel(HeadArgs a) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= *a.d_m) return;
  const float *h = a.hid + n * a.ldh;
  const int hh = a.hh, nc = a.classes;
  float logit[MAX_CLASSES];
  float mx = -__builtin_inff();
  for (int k = 0; k < nc; ++k) {
    float s = 0.0f;
    for (int j = 0; j < hh; ++j) s += h[j] * a.W1[k * hh + j];
    s += a.b1[k];
    logit[k] = s;
    mx = fmaxf(mx, s);
  }
  float den = 0.0f;
  float pr[MAX_CLASSES];
  for (int k = 0; k < nc; ++k) { pr[k] = expf(logit[k] - mx); den += pr[k]; }
  int arg = 0;
  float best = -1.0f;
  for (int k = 0; k < nc; ++k) {
    pr[k] = pr[k] / den;
    if (pr[k] > best) { best = pr[k]; arg = k; }      // torch.argmax: first maximal index
  }
  float s = 0.0f;
  const float *wc = a.W1 + nc * hh;
  for (int j = 0; j < hh; ++j) s += h[hh + j] * wc[j];
  s += a.b1[nc];
  const float conf = 1.0f / (1.0f + expf(-s));
  float corr = 0.0f;
  if (a.has_corr) {
    const float *wr = wc + hh;
    float t = 0.0f;
    for (int j = 0; j < hh; ++j) t += h[2 * hh + j] * wr[j];
    corr = t + a.b1[nc + 1];
  }
  if (a.o.class_logits) for (int k = 0; k < nc; ++k) a.o.class_logits[n * nc + k] = logit[k];
  if (a.o.class_probs) for (int k = 0; k < nc; ++k) a.o.class_probs[n * nc + k] = pr[k];
  if (a.o.predicted_class) a.o.predicted_class[n] = arg;
  if (a.o.confidence) a.o.confidence[n] = conf;
  if (a.o.correction && a.has_corr) a.o.correction[n] = corr;
  int action = 0;
  if (arg == 2 && conf > a.thr_auto) action = 1;      // CLASS_NOISE = 2 (gnn.py:279,439-441)
  if (conf < a.thr_review) action = 2;                // overrides (gnn.py:444-445)
  if (a.o.action) a.o.action[n] = action;
  if (a.o.needs_review) a.o.needs_review[n] = action == 2;
  if (a.o.auto_correct) a.o.auto_correct[n] = action == 1;
}

// fast path: compile-time class count and hidden width, float4 row loads, everything in registers
template <int NCLS, int HH>
__global__ __launch_bounds__(256) void heads_final_fast_kernel(HeadArgs a) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= *a.d_m) return;
  const float4 *h4 = reinterpret_cast<const float4 *>(a.hid + n * a.ldh);
  float logit[NCLS];
#pragma unroll
  for (int k = 0; k < NCLS; ++k) logit[k] = 0.0f;
  float sc = 0.0f, sr = 0.0f;
#pragma unroll
  for (int j = 0; j < HH / 4; ++j) {
    const float4 x = h4[j];
#pragma unroll
    for (int k = 0; k < NCLS; ++k) {
      const float *w = a.W1 + k * HH + 4 * j;
      logit[k] += x.x * w[0] + x.y * w[1] + x.z * w[2] + x.w * w[3];
    }
  }
#pragma unroll
  for (int j = 0; j < HH / 4; ++j) {
    const float4 x = h4[HH / 4 + j];
    const float *w = a.W1 + NCLS * HH + 4 * j;
    sc += x.x * w[0] + x.y * w[1] + x.z * w[2] + x.w * w[3];
  }
  if (a.has_corr) {
#pragma unroll
    for (int j = 0; j < HH / 4; ++j) {
      const float4 x = h4[2 * (HH / 4) + j];
      const float *w = a.W1 + (NCLS + 1) * HH + 4 * j;
      sr += x.x * w[0] + x.y * w[1] + x.z * w[2] + x.w * w[3];
    }
  }
  float mx = -__builtin_inff();
#pragma unroll
  for (int k = 0; k < NCLS; ++k) { logit[k] += a.b1[k]; mx = fmaxf(mx, logit[k]); }
  float pr[NCLS], den = 0.0f;
#pragma unroll
  for (int k = 0; k < NCLS; ++k) { pr[k] = expf(logit[k] - mx); den += pr[k]; }
  int arg = 0;
  float best = -1.0f;
#pragma unroll
  for (int k = 0; k < NCLS; ++k) {
    pr[k] = pr[k] / den;
    if (pr[k] > best) { best = pr[k]; arg = k; }
  }
  const float conf = 1.0f / (1.0f + expf(-(sc + a.b1[NCLS])));
  const float corr = sr + a.b1[NCLS + 1];
#pragma unroll
  for (int k = 0; k < NCLS; ++k) {
    if (a.o.class_logits) a.o.class_logits[n * NCLS + k] = logit[k];
    if (a.o.class_probs) a.o.class_probs[n * NCLS + k] = pr[k];
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
}

int launch_heads_final(bgnn_ctx *ctx, const bgnn_model *m, const float *hid, int ldh, const int64_t *d_m,
                       int64_t max_rows, float thr_auto, float thr_review, const bgnn_outputs *o) {
  if (max_rows <= 0) return BGNN_OK;
  ProfScope ps(ctx, BGNN_K_HEADS);
  BGNN_REQUIRE(m->desc.num_classes <= MAX_CLASSES, "num_classes > %d unsupported", MAX_CLASSES);
  HeadArgs a{};
  a.hid = hid; a.ldh = ldh; a.W1 = m->hd_W1; a.b1 = m->hd_b1; a.d_m = d_m;
  a.hh = m->desc.hidden / 2; a.classes = m->desc.num_classes; a.has_corr = m->desc.predict_correction;
  a.thr_auto = thr_auto; a.thr_review = thr_review; a.o = *o;
  const dim3 grid((unsigned)((max_rows + 255) / 256)), block(256);
  if (a.classes == 3 && a.hh == 32 && ldh % 4 == 0)
    hipLaunchKernelGGL((heads_final_fast_kernel<3, 32>), grid, block, 0, ctx->stream, a);
  else if (a.classes == 3 && a.hh == 16 && ldh % 4 == 0)
    hipLaunchKernelGGL((heads_final_fast_kernel<3, 16>), grid, block, 0, ctx->stream, a);
  else if (a.classes == 3 && a.hh == 64 && ldh % 4 == 0)          // hidden 128 (and the widths padded to it): the runtime-width kernel took 25 ms per 8.4 M nodes
    hipLaunchKernelGGL((heads_final_fast_kernel<3, 64>), grid, block, 0, ctx->stream, a);
  else
    hipLaunchKernelGGL(heads_final_kernel, grid, block, 0, ctx->stream, a);
  BGNN_HIP_CHECK(hipGetLastError());
  return BGNN_OK;
}

}  // namespace bgnn
