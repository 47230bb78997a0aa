// C ABI of libbgnn_hip.so (declared in include/bgnn.h): contexts, weights, graph handles and the
// forward / fused-inference orchestration.  Host code only; kernels live in the other TUs.
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>

#include <algorithm>
#include <cmath>

#include "bgnn_internal.h"

namespace bgnn {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- pool -----------------------------------------------------------------------------------
int DevPool::alloc(size_t bytes, void **out) {
  if (bytes == 0) bytes = 256;
  bytes = (bytes + 255) & ~(size_t)255;
  auto it = free_blocks.lower_bound(bytes);
  if (it != free_blocks.end() && it->first <= bytes + bytes / 4 + 4096) {
    *out = it->second;
    live[*out] = it->first;
    free_blocks.erase(it);
    return BGNN_OK;
  }
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    trim();
    e = hipMalloc(&p, bytes);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error("device allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    return BGNN_ERR_NOMEM;
  }
  total_bytes += bytes;
  live[p] = bytes;
  *out = p;
  return BGNN_OK;
}

void DevPool::release(void *p) {
  if (!p) return;
  auto it = live.find(p);
  if (it == live.end()) return;
  free_blocks.emplace(it->second, p);
  live.erase(it);
}

void DevPool::trim() {
  for (auto &kv : free_blocks) {
    (void)hipFree(kv.second);
    total_bytes -= kv.first;
  }
  free_blocks.clear();
}

int ctx_workspace(bgnn_ctx *ctx, int slot, size_t bytes, void **out) {
  if (ctx->ws_bytes[slot] < bytes) {
    if (ctx->ws[slot]) {
      BGNN_HIP_CHECK(hipStreamSynchronize(ctx->stream));
      BGNN_HIP_CHECK(hipFree(ctx->ws[slot]));
      ctx->ws[slot] = nullptr;
      ctx->ws_bytes[slot] = 0;
    }
    size_t want = bytes + bytes / 8;
    hipError_t e = hipMalloc(&ctx->ws[slot], want);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      ctx->pool.trim();
      want = bytes;
      e = hipMalloc(&ctx->ws[slot], want);
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      set_error("workspace allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
      return BGNN_ERR_NOMEM;
    }
    ctx->ws_bytes[slot] = want;
  }
  *out = ctx->ws[slot];
  return BGNN_OK;
}

int ctx_upload(bgnn_ctx *ctx, const void *host, size_t bytes, void *dev) {
  if (bytes == 0) return BGNN_OK;
  bgnn_ctx::Staging *slot = nullptr;
  for (auto &st : ctx->staging) {
    if (st.cap < bytes) continue;
    if (st.in_flight && hipEventQuery(st.ev) != hipSuccess) continue;
    slot = &st;
    break;
  }
  (void)hipGetLastError();                            // hipEventQuery reports "not ready" as an error code
  if (!slot) {
    bgnn_ctx::Staging st{nullptr, std::max<size_t>(bytes, 64 * 1024), nullptr, false};
    BGNN_HIP_CHECK(hipHostMalloc(&st.p, st.cap, hipHostMallocDefault));
    BGNN_HIP_CHECK(hipEventCreateWithFlags(&st.ev, hipEventDisableTiming));
    ctx->staging.push_back(st);
    slot = &ctx->staging.back();
  }
  memcpy(slot->p, host, bytes);
  BGNN_HIP_CHECK(hipMemcpyAsync(dev, slot->p, bytes, hipMemcpyHostToDevice, ctx->stream));
  BGNN_HIP_CHECK(hipEventRecord(slot->ev, ctx->stream));
  slot->in_flight = true;
  return BGNN_OK;
}

ProfScope::ProfScope(bgnn_ctx *c, int kernel) : ctx(c), idx(-1) {
  if (!(c->prof_mask & (1u << kernel))) return;
  ProfRecord r;
  r.kernel = kernel;
  hipEvent_t ev[2];
  for (int i = 0; i < 2; ++i) {
    if (!c->event_pool.empty()) { ev[i] = c->event_pool.back(); c->event_pool.pop_back(); }
    else if (hipEventCreate(&ev[i]) != hipSuccess) return;
  }
  r.start = ev[0]; r.stop = ev[1];
  (void)hipEventRecord(r.start, c->stream);
  c->prof_records.push_back(r);
  idx = (int)c->prof_records.size() - 1;
}

ProfScope::~ProfScope() {
  if (idx >= 0) (void)hipEventRecord(ctx->prof_records[idx].stop, ctx->stream);
}

// x [n][in] -> x8 [n][8], zero padded (the node-feature layout of the graph build)
__global__ void pad_rows8_kernel(const float *x, int in, float *x8, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n * 8) x8[i] = (i & 7) < in ? x[(i >> 3) * in + (i & 7)] : 0.0f;
}

}  // namespace bgnn

using namespace bgnn;

extern "C" {

int bgnn_abi_version(void) { return BGNN_ABI_VERSION; }
const char *bgnn_last_error(void) { return g_err; }

// ---- context ----------------------------------------------------------------------------------
int bgnn_ctx_create(int device, void *stream, bgnn_ctx **out) {
  BGNN_REQUIRE(out != nullptr, "bgnn_ctx_create: out is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    set_error("no HIP device available (%s)", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    return BGNN_ERR_HIP;
  }
  BGNN_REQUIRE(device >= 0 && device < n, "bgnn_ctx_create: device %d out of range (have %d)", device, n);
  BGNN_HIP_CHECK(hipSetDevice(device));
  hipDeviceProp_t prop;
  BGNN_HIP_CHECK(hipGetDeviceProperties(&prop, device));
  bgnn_ctx *c = new bgnn_ctx();
  c->device = device;
  c->num_cus = prop.multiProcessorCount;
  if (stream) {
    c->stream = (hipStream_t)stream;
    c->owns_stream = false;
  } else {
    hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (se != hipSuccess) {
      delete c;
      set_error("hipStreamCreate failed: %s", hipGetErrorString(se));
      return BGNN_ERR_HIP;
    }
    c->owns_stream = true;
  }
  if (hipMalloc((void **)&c->zero_page, 16384) != hipSuccess || hipMemset(c->zero_page, 0, 16384) != hipSuccess) {
    set_error("zero page allocation failed");
    delete c;
    return BGNN_ERR_NOMEM;
  }
  c->stamps = reinterpret_cast<unsigned long long *>(c->zero_page + 1024);
  {   // defaults from the environment, read once per context
    auto env_int = [](const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; };
    BgnnOpts &o = c->opts;
    o.matrix_path = getenv("BGNN_BF16") ? 3 : getenv("BGNN_SPLIT_F16") ? 2 : getenv("BGNN_SPLIT_BF16") ? 1 : 0;
    o.fused = getenv("BGNN_NO_FUSED") ? 0 : 1;
    o.fold_extractor = getenv("BGNN_NO_FOLD") ? 0 : 1;
    o.ragged_atlas = getenv("BGNN_NO_ATLAS") ? 0 : 1;
    o.fused_front = getenv("BGNN_NO_FUSED_FRONT") ? 0 : 1;
    o.fused_persistent = getenv("BGNN_PERSISTENT") ? 1 : 0;
    o.bf16_two_phase = getenv("BGNN_NO_TWO_PHASE") ? 0 : env_int("BGNN_TWO_PHASE", 1);
    o.bf16_layer0_af = getenv("BGNN_NO_LAYER0_AF") ? 0 : 1;
    o.stats_narrow = env_int("BGNN_STATS_NARROW", -1);
    o.fused_lds_pad_kb = env_int("BGNN_FUSED_LDS_PAD", 0);
    o.diag_mask = env_int("BGNN_FUSED_DBG", 0);
    o.diag_stamps = getenv("BGNN_FUSED_STAMPS") ? 1 : 0;
    o.gemm_waves = env_int("BGNN_GEMM_WAVES", 8);
    o.gemm_diag = env_int("BGNN_GEMM_DBG", 0);
    o.gemm_no_wres = getenv("BGNN_NO_WRES") ? 1 : 0;
    o.gemm_pair_major = getenv("BGNN_NO_PAIR_MAJOR") ? 0 : 1;
  }
  *out = c;
  return BGNN_OK;
}

static int *option_slot(bgnn_ctx *ctx, const char *name) {
  BgnnOpts &o = ctx->opts;
  struct { const char *n; int *p; } tab[] = {
      {"matrix_path", &o.matrix_path}, {"fused", &o.fused}, {"fold_extractor", &o.fold_extractor}, {"ragged_atlas", &o.ragged_atlas}, {"features_tiled", &o.features_tiled}, {"fused_front", &o.fused_front}, {"fused_persistent", &o.fused_persistent}, {"bf16_two_phase", &o.bf16_two_phase}, {"bf16_layer0_af", &o.bf16_layer0_af}, {"stats_narrow", &o.stats_narrow},
      {"fused_lds_pad_kb", &o.fused_lds_pad_kb},
      {"diag_mask", &o.diag_mask}, {"diag_stamps", &o.diag_stamps}, {"gemm_waves", &o.gemm_waves},
      {"gemm_diag", &o.gemm_diag}, {"gemm_no_wres", &o.gemm_no_wres}, {"gemm_pair_major", &o.gemm_pair_major}};
  for (auto &t : tab) if (strcmp(t.n, name) == 0) return t.p;
  return nullptr;
}

int bgnn_ctx_set_option(bgnn_ctx *ctx, const char *name, int value) {
  BGNN_REQUIRE(ctx && name, "bgnn_ctx_set_option: NULL argument");
  int *p = option_slot(ctx, name);
  BGNN_REQUIRE(p, "bgnn_ctx_set_option: unknown option '%s'", name);
  if (p == &ctx->opts.matrix_path) BGNN_REQUIRE(value >= 0 && value <= 3, "matrix_path=%d (0 exact f32, 1 bf16x3, 2 fp16x3, 3 bf16 storage)", value);
  if ((p == &ctx->opts.diag_mask || p == &ctx->opts.diag_stamps || p == &ctx->opts.gemm_diag) && value != 0)
    BGNN_REQUIRE(BGNN_DIAG, "option '%s' needs the diagnostic build of the library (python __graft_entry__.py --diag)", name);
  *p = value;
  return BGNN_OK;
}

int bgnn_ctx_get_option(bgnn_ctx *ctx, const char *name, int *value) {
  BGNN_REQUIRE(ctx && name && value, "bgnn_ctx_get_option: NULL argument");
  int *p = option_slot(ctx, name);
  BGNN_REQUIRE(p, "bgnn_ctx_get_option: unknown option '%s'", name);
  *value = *p;
  return BGNN_OK;
}

// diagnostic (not part of the documented ABI): read and clear the fused kernel's phase counters
// (out: 64 counters -- 0..15 the one-block-per-workgroup kernels together, 16..31 the persistent kernel, 32..47 the 256 -> 64
//  instance, 48..63 the heads instance)
int bgnn_debug_stamps(bgnn_ctx *ctx, unsigned long long *out32) {
  if (!ctx || !out32) return BGNN_ERR_INVALID;
  (void)hipStreamSynchronize(ctx->stream);
  if (hipMemcpy(out32, ctx->stamps, 512, hipMemcpyDeviceToHost) != hipSuccess) return BGNN_ERR_HIP;
  (void)hipMemset(ctx->stamps, 0, 512);
  return BGNN_OK;
}

static void graph_free(bgnn_graph *g);

int bgnn_ctx_destroy(bgnn_ctx *ctx) {
  if (!ctx) return BGNN_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  // graphs still alive on this context go with it: their handles are invalid from here on (bgnn.h: bgnn_ctx_destroy)
  while (!ctx->live_graphs.empty()) graph_free(*ctx->live_graphs.begin());
  for (auto &r : ctx->prof_records) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
  for (auto &e : ctx->event_pool) (void)hipEventDestroy(e);
  for (auto &st : ctx->staging) { (void)hipEventDestroy(st.ev); (void)hipHostFree(st.p); }
  for (int i = 0; i < 6; ++i) if (ctx->ws[i]) (void)hipFree(ctx->ws[i]);
  for (auto &e : ctx->table_cache) { (void)hipFree(e.d_tiles); (void)hipFree(e.d_items); }
  if (ctx->zero_page) (void)hipFree(ctx->zero_page);
  ctx->pool.trim();
  for (auto &kv : ctx->pool.live) (void)hipFree(kv.first);
  if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return BGNN_OK;
}

int bgnn_ctx_synchronize(bgnn_ctx *ctx) {
  BGNN_REQUIRE(ctx, "ctx is NULL");
  BGNN_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return BGNN_OK;
}

void *bgnn_ctx_stream(bgnn_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int bgnn_ctx_profile(bgnn_ctx *ctx, uint32_t kernel_mask) {
  BGNN_REQUIRE(ctx, "ctx is NULL");
  ctx->prof_mask = kernel_mask;
  return BGNN_OK;
}

int bgnn_ctx_profile_read(bgnn_ctx *ctx, double *ms, int64_t *launches) {
  BGNN_REQUIRE(ctx && ms && launches, "bgnn_ctx_profile_read: NULL argument");
  BGNN_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  for (auto &r : ctx->prof_records) {
    float t = 0.f;
    BGNN_HIP_CHECK(hipEventElapsedTime(&t, r.start, r.stop));
    ms[r.kernel] += (double)t;
    launches[r.kernel] += 1;
    ctx->event_pool.push_back(r.start);
    ctx->event_pool.push_back(r.stop);
  }
  ctx->prof_records.clear();
  return BGNN_OK;
}

// ---- model ------------------------------------------------------------------------------------
static int head_count(const bgnn_model_desc *d) { return d->predict_correction ? 3 : 2; }

size_t bgnn_model_weight_count(const bgnn_model_desc *d) {
  if (!d) return 0;
  const size_t hid = d->hidden, in = d->in_channels, hh = hid / 2;
  size_t n = hid * in + hid + hid * hid + hid;
  for (int l = 0; l < d->num_layers && d->gnn_type != BGNN_GNN_GAT; ++l) {
    if (d->gnn_type == BGNN_GNN_GCN) n += hid * hid + hid;
    else if (d->gnn_type == BGNN_GNN_SAGE) n += 2 * hid * hid + hid;
    else n += 2 * (hid * hid + hid);                              // GIN
    n += 4 * hid;                                                 // BatchNorm
  }
  for (int l = 0; l < d->num_layers && d->gnn_type == BGNN_GNN_GAT; ++l) {
    const bool last = l == d->num_layers - 1;
    const size_t H = last ? 1 : d->heads;
    const size_t D = l == 0 ? hid : hid * d->heads;
    const size_t HC = H * hid, W = last ? hid : HC;
    n += HC * D + 3 * HC + HC * d->edge_dim + W + 4 * W;
  }
  n += hh * hid + hh + (size_t)d->num_classes * hh + d->num_classes;
  n += hh * hid + hh + hh + 1;
  if (d->predict_correction) n += hh * hid + hh + hh + 1;
  return n;
}

// ---- bf16x3 operand split (opt-in matrix path) --------------------------------------------------------------
// w = hi + lo + O(2^-16 |w|) with hi = bf16(w), lo = bf16(w - hi), round to nearest even.  The image replaces Wt
// [D][NC] float32 byte for byte: per 16-row half-chunk, per 32-column tile t, per part p (hi, lo), one 1-KiB block in
// the lane order of v_mfma_f32_32x32x16_bf16's A operand: [k-group 2][column m 32][k 8] bf16.
static inline uint16_t bf16_rne(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

static inline uint16_t f16_rne(float f) {               // float32 -> IEEE half, round to nearest even, overflow -> inf
  uint32_t u; memcpy(&u, &f, 4);
  const uint32_t sign = (u >> 16) & 0x8000u;
  const int32_t e = (int32_t)((u >> 23) & 0xff) - 127 + 15;
  uint32_t m = u & 0x7fffffu;
  if (((u >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u : 0));
  if (e >= 31) return (uint16_t)(sign | 0x7c00u);
  if (e <= 0) {                                            // subnormal half (or zero)
    if (e < -10) return (uint16_t)sign;
    m |= 0x800000u;
    const int shift = 14 - e;                              // 24-bit significand -> 10 bits at exponent 2^-14
    const uint32_t half = m >> shift, rem = m & ((1u << shift) - 1), mid = 1u << (shift - 1);
    return (uint16_t)(sign | (half + ((rem > mid || (rem == mid && (half & 1))) ? 1 : 0)));
  }
  const uint32_t half = ((uint32_t)e << 10) | (m >> 13), rem = m & 0x1fffu;
  return (uint16_t)(sign | (half + ((rem > 0x1000u || (rem == 0x1000u && (half & 1))) ? 1 : 0)));
}
static inline float f16_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ffu;
  uint32_t u;
  if (e == 0) {
    if (m == 0) u = sign;
    else { int k = 0; uint32_t mm = m; while (!(mm & 0x400u)) { mm <<= 1; ++k; } u = sign | ((uint32_t)(113 - k) << 23) | ((mm & 0x3ffu) << 13); }
  } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
  else u = sign | ((e + 112) << 23) | (m << 13);
  float f; memcpy(&f, &u, 4); return f;
}

// float16 images hold W * 2^S, S chosen so that the largest |w| lands in [2^12, 2^13): the lo part of an element is then ~2^-11 of it
// and NORMAL in float16 for everything within 2^14 of the largest weight -- unscaled, the lo parts of glorot-sized weights (|w| <=
// 0.14, lo <= 6.7e-5) sat at float16's smallest normal and were carried with an absolute step of 2^-24, i.e. ~21 bits of W: that,
// not the dropped lo x lo term, was what put fp16x3 2.4x farther from the float64 forward than the exact path (profiles/NOTES_r05.md).
// The kernels multiply their accumulators by 2^-S (*inv_scale; exact) before the epilogue.  Returns false when a weight is beyond float16's range.
static bool pack_split(const float *Wt, int D, int NC, float *dst_as_float, bool f16, float *inv_scale = nullptr) {
  float sc = 1.0f;
  if (f16) {
    float amax = 0.0f;
    for (size_t i = 0; i < (size_t)D * NC; ++i) {
      if (!(std::fabs(Wt[i]) < 65504.0f)) return false;   // (a weight that large also drives the ACTIVATIONS out of float16's range: bf16 split instead)
      amax = std::max(amax, std::fabs(Wt[i]));
    }
    if (amax > 0.0f) {
      int e;
      std::frexp(amax, &e);                              // amax = m 2^e, m in [0.5, 1)
      const int S = std::max(-100, std::min(100, 13 - e));
      sc = std::ldexp(1.0f, S);
    }
  }
  if (inv_scale) *inv_scale = 1.0f / sc;
  uint16_t *dst = reinterpret_cast<uint16_t *>(dst_as_float);
  const int NT = NC / 32;
  for (int hc = 0; hc < D / 16; ++hc)
    for (int t = 0; t < NT; ++t)
      for (int part = 0; part < 2; ++part)
        for (int kg = 0; kg < 2; ++kg)
          for (int m = 0; m < 32; ++m)
            for (int i = 0; i < 8; ++i) {
              const float w = Wt[(size_t)(hc * 16 + kg * 8 + i) * NC + t * 32 + m] * sc;
              const uint16_t hi = f16 ? f16_rne(w) : bf16_rne(w);
              const uint16_t v = part == 0 ? hi : f16 ? f16_rne(w - f16_to_f32(hi)) : bf16_rne(w - bf16_to_f32(hi));
              dst[((((size_t)hc * NT + t) * 2 + part) * 2 + kg) * 256 + m * 8 + i] = v;
            }
  return true;
}

// bf16 (hi only) image for the bf16 storage path: [D/16 half-chunks][NC/32 tiles][1 KiB = k-group 2 x column 32 x k 8] in MFMA
// A-fragment lane order.  Every GEMM of that path takes an MFMA RESULT tile as its B operand (the aggregation's in the fused layer
// kernel -- gat_layer_fused.hip AggWindow --, extractor layer 1's in the lin_0 GEMM), so element i of lane half kg is
// k = 8 (i >> 2) + 4 kg + (i & 3) of the 16-k step, not 8 kg + i
static void pack_bf16_image_accop(const float *Wt, int D, int NC, float *dst_as_float) {
  uint16_t *dst = reinterpret_cast<uint16_t *>(dst_as_float);
  const int NT = NC / 32;
  for (int hc = 0; hc < D / 16; ++hc)
    for (int t = 0; t < NT; ++t)
      for (int kg = 0; kg < 2; ++kg)
        for (int m = 0; m < 32; ++m)
          for (int i = 0; i < 8; ++i)
            dst[(((size_t)hc * NT + t) * 2 + kg) * 256 + m * 8 + i] =
                bf16_rne(Wt[(size_t)(hc * 16 + 8 * (i >> 2) + 4 * kg + (i & 3)) * NC + t * 32 + m]);
}

// Alpha tile of the bf16 front GEMM (gemm_f32.hip, AMF): the attention dots alpha_src[hd] = sum_c Y[hd C + c] att_src[hd C + c]
// with Y = x W + b are x (W att) + b att.  32 weight columns behind the W image, packed like one more tile: column hd = the hi
// bf16 part of sum_c W_bf16[k][hd C + c] att_src[hd C + c], 4 + hd the same for att_dst, 8 + hd / 12 + hd the lo parts (hi + lo:
// 16 mantissa bits; W_bf16 = the rounded weights the GEMM itself multiplies by), the rest zero; then 8 floats: b att per head.
static void pack_alpha_tile(const float *Wt, const float *bias, const float *att_src, const float *att_dst, int D, int H, int C,
                            float *dst) {
  std::vector<float> Wa((size_t)D * 32, 0.0f);
  for (int k = 0; k < D; ++k)
    for (int hd = 0; hd < H; ++hd) {
      double s = 0.0, d = 0.0;
      for (int c = 0; c < C; ++c) {
        const double w = (double)bf16_to_f32(bf16_rne(Wt[(size_t)k * H * C + hd * C + c]));
        s += w * (double)att_src[hd * C + c];
        d += w * (double)att_dst[hd * C + c];
      }
      const float fs = (float)s, fd = (float)d;
      const float hs = bf16_to_f32(bf16_rne(fs)), hd_ = bf16_to_f32(bf16_rne(fd));
      Wa[(size_t)k * 32 + hd] = hs;       Wa[(size_t)k * 32 + 8 + hd] = bf16_to_f32(bf16_rne(fs - hs));
      Wa[(size_t)k * 32 + 4 + hd] = hd_;  Wa[(size_t)k * 32 + 12 + hd] = bf16_to_f32(bf16_rne(fd - hd_));
    }
  pack_bf16_image_accop(Wa.data(), D, 32, dst);
  float *cb = dst + (size_t)D / 16 * 256;
  for (int hd = 0; hd < 8; ++hd) cb[hd] = 0.0f;
  for (int hd = 0; hd < H; ++hd) {
    double s = 0.0, d = 0.0;
    for (int c = 0; c < C; ++c) {
      s += (double)(bias ? bias[hd * C + c] : 0.0f) * (double)att_src[hd * C + c];
      d += (double)(bias ? bias[hd * C + c] : 0.0f) * (double)att_dst[hd * C + c];
    }
    cb[hd] = (float)s; cb[4 + hd] = (float)d;
  }
}

// column-permuted f32 image for the fused exact-f32 kernel: column 32 t + r of a row goes to (t / TG) * 32 TG + r * TG + t % TG,
// TG = 4 / 2 / 1 tiles per LDS read (gat_layer_fused.hip: WTileGroup)
// (tg > 0 forces the group width: the lin_0 GEMM's pair-major form reads TWO tiles per ds_read_b64, gemm_f32.hip PM)
// [D][NC] -> [NC / 256][D][256]: the 256-column blocks of a wide layer, each a contiguous image for the generic GEMM
static void pack_col_blocks(const float *Wt, int D, int NC, float *dst) {
  for (int b = 0; b < NC / 256; ++b)
    for (int k = 0; k < D; ++k)
      for (int c = 0; c < 256; ++c) dst[((size_t)b * D + k) * 256 + c] = Wt[(size_t)k * NC + b * 256 + c];
}

static void pack_tilegroup_image(const float *Wt, int D, int NC, float *dst, int tg = 0) {
  const int NT = NC / 32, TG = tg > 0 ? tg : NT % 4 == 0 ? 4 : NT % 2 == 0 ? 2 : 1;
  for (int k = 0; k < D; ++k)
    for (int t = 0; t < NT; ++t)
      for (int r = 0; r < 32; ++r)
        dst[(size_t)k * NC + (t / TG) * 32 * TG + r * TG + t % TG] = Wt[(size_t)k * NC + t * 32 + r];
}

// ---- model widths the kernels have no instance for: zero padding -------------------------------------------------------------
// The kernels exist for hidden 32 / 64 / 128 and power-of-two head counts.  Any other width the reference's config allows
// (config/config.py:43-45: any gnn_hidden_channels / gnn_heads) is embedded in the next supported one: channel c of head h goes to
// column h * Cp + c, everything else is zero weight, zero bias, BatchNorm (weight 1, bias 0, mean 0, var 1).  A padded channel is
// then exactly 0.0 at every stage (Linear: 0, ReLU: 0, GATConv: alpha * 0 summed, + bias 0, BatchNorm: (0 - 0) s + 0), a padded
// head's attention logits are all leaky_relu(0) (a uniform softmax over zeros), and a real channel only ever sees added +0.0 terms:
// the results of the logical model, in another summation grouping.  Input: the flat blob in bgnn_model_weight_count's order.
static inline int pad_hidden(int c) { return c <= 32 ? 32 : c <= 64 ? 64 : 128; }
static inline int pad_heads(int h) { int p = 1; while (p < h) p <<= 1; return p; }

static void pad_model_weights(const bgnn_model_desc *d, const float *w, bgnn_model_desc *dp, std::vector<float> &out) {
  const bool gat = d->gnn_type == BGNN_GNN_GAT;
  *dp = *d;
  dp->hidden = pad_hidden(d->hidden);
  if (gat) dp->heads = pad_heads(d->heads);
  const int C = d->hidden, Cp = dp->hidden, Hh = d->heads, in = d->in_channels, hh = C / 2, hhp = Cp / 2, L = d->num_layers, ED = d->edge_dim;
  out.assign(bgnn_model_weight_count(dp), 0.0f);
  const float *p = w;
  float *q = out.data();
  // index maps: a plain width-C vector, and the concatenation of H heads of C channels
  auto ident = [](int n) { std::vector<int> m(n); for (int i = 0; i < n; ++i) m[i] = i; return m; };
  auto headmap = [&](int H) { std::vector<int> m((size_t)H * C); for (int h = 0; h < H; ++h) for (int c = 0; c < C; ++c) m[(size_t)h * C + c] = h * Cp + c; return m; };
  // matrix [rows][cols] (torch Linear weight: [out][in]) -> [rows_p][cols_p], vector likewise; `fill` for the pad entries of a vector
  auto mat = [&](const std::vector<int> &rm, int rows_p, const std::vector<int> &cm, int cols_p) {
    for (size_t r = 0; r < rm.size(); ++r)
      for (size_t c = 0; c < cm.size(); ++c) q[(size_t)rm[r] * cols_p + cm[c]] = p[r * cm.size() + c];
    p += rm.size() * cm.size(); q += (size_t)rows_p * cols_p;
  };
  auto vec = [&](const std::vector<int> &m, int n_p, float fill = 0.0f) {
    for (int i = 0; i < n_p; ++i) q[i] = fill;
    for (size_t i = 0; i < m.size(); ++i) q[m[i]] = p[i];
    p += m.size(); q += n_p;
  };
  auto batch_norm = [&](const std::vector<int> &m, int n_p) { vec(m, n_p, 1.0f); vec(m, n_p); vec(m, n_p); vec(m, n_p, 1.0f); };   // w, b, mean, var
  const std::vector<int> mC = ident(C), mIn = ident(in), mHh = ident(hh), mED = ident(ED);
  mat(mC, Cp, mIn, in); vec(mC, Cp);
  mat(mC, Cp, mC, Cp); vec(mC, Cp);
  for (int l = 0; l < L && !gat; ++l) {
    mat(mC, Cp, mC, Cp); vec(mC, Cp);                                                  // GCN: lin, bias; SAGE: lin_l, bias; GIN: nn.0
    if (d->gnn_type == BGNN_GNN_SAGE) mat(mC, Cp, mC, Cp);                             // lin_r
    if (d->gnn_type == BGNN_GNN_GIN) { mat(mC, Cp, mC, Cp); vec(mC, Cp); }             // nn.2
    batch_norm(mC, Cp);
  }
  for (int l = 0; l < L && gat; ++l) {
    const bool last = l == L - 1;
    const int H = last ? 1 : Hh, Hp = last ? 1 : dp->heads;
    const std::vector<int> mOut = headmap(H), mInL = l == 0 ? mC : headmap(Hh);
    const int outp = Hp * Cp, inp = l == 0 ? Cp : dp->heads * Cp;
    mat(mOut, outp, mInL, inp);                                                        // lin.weight [HC][D]
    vec(mOut, outp); vec(mOut, outp); vec(mOut, outp);                                 // att_src, att_dst, att_edge
    mat(mOut, outp, mED, ED);                                                          // lin_edge.weight [HC][ED]
    const std::vector<int> &mW = last ? mC : mOut;                                     // (last layer: mean over its one head -> [C])
    const int wp = last ? Cp : outp;
    vec(mW, wp);                                                                       // bias
    batch_norm(mW, wp);
  }
  const int nh = head_count(d);
  for (int k = 0; k < nh; ++k) {
    const int nout = k == 0 ? d->num_classes : 1;
    mat(mHh, hhp, mC, Cp); vec(mHh, hhp);                                              // mlp.0
    mat(ident(nout), nout, mHh, hhp); vec(ident(nout), nout);                          // mlp.3
  }
}

__global__ void copy_cols_kernel(const float *src, int src_stride, float *dst, int dst_stride, int n_copy, const int64_t *d_m) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= *d_m * dst_stride) return;
  const int64_t r = i / dst_stride;
  const int c = (int)(i - r * dst_stride);
  dst[i] = c < n_copy ? src[r * src_stride + c] : 0.0f;
}

// rows [*d_m][src_stride] -> [*d_m][dst_stride]: the first n_copy columns, the rest of a destination row zero
static int launch_copy_cols(bgnn_ctx *ctx, const float *src, int src_stride, float *dst, int dst_stride, int n_copy, const int64_t *d_m,
                            int64_t rows_cap) {
  const int64_t n = rows_cap * dst_stride;
  if (n <= 0) return BGNN_OK;
  hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, src, src_stride, dst, dst_stride,
                     n_copy, d_m);
  BGNN_HIP_CHECK(hipGetLastError());
  return BGNN_OK;
}

static int model_create_native(bgnn_ctx *ctx, const bgnn_model_desc *d, const float *w, size_t n_weights, bgnn_model **out);

int bgnn_model_create(bgnn_ctx *ctx, const bgnn_model_desc *d_in, const float *w, size_t n_weights, bgnn_model **out) {
  BGNN_REQUIRE(ctx && d_in && w && out, "bgnn_model_create: NULL argument");
  bgnn_model_desc dl = *d_in;                                  // the LOGICAL model
  BGNN_REQUIRE(dl.gnn_type >= BGNN_GNN_GAT && dl.gnn_type <= BGNN_GNN_GIN, "gnn_type=%d unknown", dl.gnn_type);
  const bool gat = dl.gnn_type == BGNN_GNN_GAT;
  if (!gat) dl.heads = 1;                                      // (`heads` only shapes a GAT backbone: models/gnn.py:125-143)
  BGNN_REQUIRE(dl.hidden >= 2 && dl.hidden <= 128, "hidden_channels=%d unsupported (2..128)", dl.hidden);
  BGNN_REQUIRE(dl.heads >= 1 && dl.heads <= 256 && pad_heads(dl.heads) * pad_hidden(dl.hidden) <= 512,
               "heads=%d x hidden_channels=%d unsupported: the layer is laid out as %d heads of %d channels (next power of two x next of "
               "32 / 64 / 128), which must stay within 512 columns", dl.heads, dl.hidden, pad_heads(dl.heads), pad_hidden(dl.hidden));
  BGNN_REQUIRE(dl.in_channels >= 1 && dl.in_channels <= 8, "in_channels=%d unsupported (1..8)", dl.in_channels);
  BGNN_REQUIRE(dl.num_layers >= 1 && dl.num_layers <= 64, "num_gnn_layers=%d unsupported", dl.num_layers);
  BGNN_REQUIRE(!gat || (dl.edge_dim >= 1 && dl.edge_dim <= 4), "edge_dim=%d unsupported (1..4)", dl.edge_dim);
  BGNN_REQUIRE(dl.num_classes >= 1 && dl.num_classes <= 16, "num_classes=%d unsupported", dl.num_classes);
  BGNN_REQUIRE(n_weights == bgnn_model_weight_count(&dl), "weight blob has %zu floats, expected %zu", n_weights,
               bgnn_model_weight_count(&dl));
  const bool padded = pad_hidden(dl.hidden) != dl.hidden || (gat && pad_heads(dl.heads) != dl.heads);
  int rc;
  if (!padded) {
    rc = model_create_native(ctx, &dl, w, n_weights, out);
  } else {
    bgnn_model_desc dp;
    std::vector<float> wp;
    pad_model_weights(&dl, w, &dp, wp);
    rc = model_create_native(ctx, &dp, wp.data(), wp.size(), out);
  }
  if (rc != BGNN_OK) return rc;
  (*out)->logical_hidden = dl.hidden; (*out)->logical_heads = dl.heads; (*out)->padded = padded;
  return BGNN_OK;
}

static int model_create_native(bgnn_ctx *ctx, const bgnn_model_desc *d, const float *w, size_t n_weights, bgnn_model **out) {
  // (the generic kernels take 32 / 64 / 128 as long as a layer stays within 256 columns -- the heads' hidden/2 has to be a multiple of
  //  16 and their three first layers side by side a multiple of 32; the fused kernels exist for hidden 64 only, the reference's
  //  default: config/config.py:41)
  BGNN_REQUIRE(d->hidden == 32 || d->hidden == 64 || d->hidden == 128, "hidden_channels=%d unsupported (32, 64 or 128)", d->hidden);
  BGNN_REQUIRE(d->in_channels >= 1 && d->in_channels <= 8, "in_channels=%d unsupported (1..8)", d->in_channels);
  BGNN_REQUIRE(d->num_layers >= 1 && d->num_layers <= 64, "num_gnn_layers=%d unsupported", d->num_layers);
  BGNN_REQUIRE(d->gnn_type >= BGNN_GNN_GAT && d->gnn_type <= BGNN_GNN_GIN, "gnn_type=%d unknown", d->gnn_type);
  const bool gat = d->gnn_type == BGNN_GNN_GAT;
  // (`heads` only shapes a GAT backbone: models/gnn.py:125-143)
  // (up to 256 columns a layer is one launch per kernel; 512 columns -- 8 heads of 64, 4 of 128 -- run the generic kernels in two
  //  256-column blocks: Wt_blk)
  BGNN_REQUIRE(!gat || (d->heads >= 1 && d->heads * d->hidden <= 512 && (d->heads & (d->heads - 1)) == 0),
               "heads=%d unsupported (power of two, heads*hidden <= 512)", d->heads);
  BGNN_REQUIRE(!gat || (d->edge_dim >= 1 && d->edge_dim <= 4), "edge_dim=%d unsupported (1..4)", d->edge_dim);
  BGNN_REQUIRE(d->num_classes >= 1 && d->num_classes <= 16, "num_classes=%d unsupported", d->num_classes);
  BGNN_REQUIRE(n_weights == bgnn_model_weight_count(d), "weight blob has %zu floats, expected %zu", n_weights,
               bgnn_model_weight_count(d));
  BGNN_HIP_CHECK(hipSetDevice(ctx->device));
  const int hid = d->hidden, in = d->in_channels, hh = hid / 2, L = d->num_layers, ED = d->edge_dim;
  const int nh = head_count(d);
  const int HT = ((nh * hh + 31) / 32) * 32;

  std::vector<float> pk;
  auto reserve = [&](size_t n) { size_t o = pk.size(); pk.resize(o + ((n + 3) & ~(size_t)3), 0.0f); return o; };
  const float *p = w;
  // feature extractor
  size_t o_fe_W0t = reserve((size_t)8 * hid), o_fe_b0 = reserve(hid);
  for (int o = 0; o < hid; ++o) for (int i = 0; i < in; ++i) pk[o_fe_W0t + (size_t)i * hid + o] = p[(size_t)o * in + i];
  p += (size_t)hid * in;
  std::copy(p, p + hid, pk.begin() + o_fe_b0); p += hid;
  size_t o_fe_W1t = reserve((size_t)hid * hid), o_fe_b1 = reserve(hid);
  for (int o = 0; o < hid; ++o) for (int i = 0; i < hid; ++i) pk[o_fe_W1t + (size_t)i * hid + o] = p[(size_t)o * hid + i];
  p += (size_t)hid * hid;
  std::copy(p, p + hid, pk.begin() + o_fe_b1); p += hid;
  struct LOff { size_t Wt, as, ad, V, sc, sh, b1, Wt2, b2, tr_bias, tr_bw, tr_bb, tr_Wt; };   // tr_*: unfolded, for bgnn_forward_train
  std::vector<LOff> lo(L);
  // BatchNorm (eval) as y = x * s + t
  auto bn_fold = [&](const float *bw, const float *bb, const float *rm, const float *rv, int c, double &sc, double &sh) {
    sc = (double)bw[c] / std::sqrt((double)rv[c] + (double)d->bn_eps);
    sh = (double)bb[c] - (double)rm[c] * sc;
  };
  for (int l = 0; l < L && !gat; ++l) {
    // every layer hid -> hid.  W^T layouts [in][out]; BatchNorm folded into the last linear map of the layer
    // (GCN: into the reduce kernel's scale / shift, because the aggregate sits between lin and bias)
    const float *W0 = p; p += (size_t)hid * hid;
    const float *b0 = nullptr, *W1 = nullptr, *b1 = nullptr;
    if (d->gnn_type == BGNN_GNN_GCN) { b0 = p; p += hid; }
    else if (d->gnn_type == BGNN_GNN_SAGE) { b0 = p; p += hid; W1 = p; p += (size_t)hid * hid; }
    else { b0 = p; p += hid; W1 = p; p += (size_t)hid * hid; b1 = p; p += hid; }
    const float *bw = p; p += hid; const float *bb = p; p += hid; const float *rm = p; p += hid; const float *rv = p; p += hid;
    {   // the unfolded last map of the layer (training-mode forward: BatchNorm statistics come from the batch)
      const float *rb = d->gnn_type == BGNN_GNN_GIN ? b1 : b0;
      lo[l].tr_bias = reserve(hid); lo[l].tr_bw = reserve(hid); lo[l].tr_bb = reserve(hid);
      std::copy(rb, rb + hid, pk.begin() + lo[l].tr_bias);
      std::copy(bw, bw + hid, pk.begin() + lo[l].tr_bw); std::copy(bb, bb + hid, pk.begin() + lo[l].tr_bb);
      lo[l].tr_Wt = 0;
      if (d->gnn_type == BGNN_GNN_SAGE) {
        lo[l].tr_Wt = reserve((size_t)2 * hid * hid);
        for (int o = 0; o < hid; ++o) for (int i = 0; i < hid; ++i) {
          pk[lo[l].tr_Wt + (size_t)i * hid + o] = W0[(size_t)o * hid + i];
          pk[lo[l].tr_Wt + (size_t)(hid + i) * hid + o] = W1[(size_t)o * hid + i];
        }
      } else if (d->gnn_type == BGNN_GNN_GIN) {
        lo[l].tr_Wt = reserve((size_t)hid * hid);
        for (int o = 0; o < hid; ++o) for (int i = 0; i < hid; ++i) pk[lo[l].tr_Wt + (size_t)i * hid + o] = W1[(size_t)o * hid + i];
      }
    }
    if (d->gnn_type == BGNN_GNN_GCN) {
      lo[l].Wt = reserve((size_t)hid * hid); lo[l].sc = reserve(hid); lo[l].sh = reserve(hid);
      for (int o = 0; o < hid; ++o) {
        for (int i = 0; i < hid; ++i) pk[lo[l].Wt + (size_t)i * hid + o] = W0[(size_t)o * hid + i];
        double sc, sh; bn_fold(bw, bb, rm, rv, o, sc, sh);
        pk[lo[l].sc + o] = (float)sc; pk[lo[l].sh + o] = (float)((double)b0[o] * sc + sh);
      }
    } else if (d->gnn_type == BGNN_GNN_SAGE) {
      lo[l].Wt = reserve((size_t)2 * hid * hid); lo[l].b2 = reserve(hid);
      for (int o = 0; o < hid; ++o) {
        double sc, sh; bn_fold(bw, bb, rm, rv, o, sc, sh);
        for (int i = 0; i < hid; ++i) {
          pk[lo[l].Wt + (size_t)i * hid + o] = (float)((double)W0[(size_t)o * hid + i] * sc);           // lin_l: mean part
          pk[lo[l].Wt + (size_t)(hid + i) * hid + o] = (float)((double)W1[(size_t)o * hid + i] * sc);     // lin_r: root part
        }
        pk[lo[l].b2 + o] = (float)((double)b0[o] * sc + sh);
      }
    } else {
      lo[l].Wt = reserve((size_t)hid * hid); lo[l].b1 = reserve(hid); lo[l].Wt2 = reserve((size_t)hid * hid); lo[l].b2 = reserve(hid);
      for (int o = 0; o < hid; ++o) {
        double sc, sh; bn_fold(bw, bb, rm, rv, o, sc, sh);
        for (int i = 0; i < hid; ++i) {
          pk[lo[l].Wt + (size_t)i * hid + o] = W0[(size_t)o * hid + i];
          pk[lo[l].Wt2 + (size_t)i * hid + o] = (float)((double)W1[(size_t)o * hid + i] * sc);
        }
        pk[lo[l].b1 + o] = b0[o];
        pk[lo[l].b2 + o] = (float)((double)b1[o] * sc + sh);
      }
    }
  }
  for (int l = 0; l < L && gat; ++l) {
    const bool last = l == L - 1;
    const int H = last ? 1 : d->heads, D = l == 0 ? hid : hid * d->heads, HC = H * hid, W = last ? hid : HC;
    lo[l].Wt = reserve((size_t)D * HC);
    for (int o = 0; o < HC; ++o) for (int i = 0; i < D; ++i) pk[lo[l].Wt + (size_t)i * HC + o] = p[(size_t)o * D + i];
    p += (size_t)HC * D;
    lo[l].as = reserve(HC); std::copy(p, p + HC, pk.begin() + lo[l].as); p += HC;
    lo[l].ad = reserve(HC); std::copy(p, p + HC, pk.begin() + lo[l].ad); p += HC;
    const float *att_edge = p; p += HC;
    const float *W_e = p; p += (size_t)HC * ED;
    lo[l].V = reserve((size_t)H * ED);
    for (int h = 0; h < H; ++h)
      for (int f = 0; f < ED; ++f) {
        double s = 0.0;
        for (int c = 0; c < hid; ++c) s += (double)att_edge[h * hid + c] * (double)W_e[(size_t)(h * hid + c) * ED + f];
        pk[lo[l].V + (size_t)h * ED + f] = (float)s;
      }
    const float *bias = p; p += W;
    const float *bw = p; p += W;
    const float *bb = p; p += W;
    const float *rm = p; p += W;
    const float *rv = p; p += W;
    lo[l].sc = reserve(W); lo[l].sh = reserve(W);
    for (int c = 0; c < W; ++c) {
      const double s = (double)bw[c] / std::sqrt((double)rv[c] + (double)d->bn_eps);
      pk[lo[l].sc + c] = (float)s;
      pk[lo[l].sh + c] = (float)(((double)bias[c] - (double)rm[c]) * s + (double)bb[c]);
    }
    lo[l].tr_bias = reserve(W); lo[l].tr_bw = reserve(W); lo[l].tr_bb = reserve(W); lo[l].tr_Wt = 0;
    std::copy(bias, bias + W, pk.begin() + lo[l].tr_bias);
    std::copy(bw, bw + W, pk.begin() + lo[l].tr_bw); std::copy(bb, bb + W, pk.begin() + lo[l].tr_bb);
  }
  const size_t o_ones = reserve(512);                     // (as wide as the widest layer: heads * hidden <= 512)
  std::fill(pk.begin() + o_ones, pk.begin() + o_ones + 512, 1.0f);
  // heads: first layers concatenated column-wise, second layers packed
  size_t o_hW0t = reserve((size_t)hid * HT), o_hb0 = reserve(HT);
  size_t o_hW1 = reserve((size_t)d->num_classes * hh + 2 * hh), o_hb1 = reserve(d->num_classes + 2);
  for (int k = 0; k < nh; ++k) {
    for (int o = 0; o < hh; ++o) for (int i = 0; i < hid; ++i) pk[o_hW0t + (size_t)i * HT + k * hh + o] = p[(size_t)o * hid + i];
    p += (size_t)hh * hid;
    std::copy(p, p + hh, pk.begin() + o_hb0 + k * hh); p += hh;
    const int nout = k == 0 ? d->num_classes : 1;
    const size_t woff = k == 0 ? 0 : (size_t)d->num_classes * hh + (size_t)(k - 1) * hh;
    std::copy(p, p + (size_t)nout * hh, pk.begin() + o_hW1 + woff); p += (size_t)nout * hh;
    const size_t boff = k == 0 ? 0 : d->num_classes + (k - 1);
    std::copy(p, p + nout, pk.begin() + o_hb1 + boff); p += nout;
  }
  // the fused heads kernel takes all of the above as ONE LDS image (a single DMA piece per workgroup): first-layer biases at 0,
  // second-layer row j at 96 + 32 j, second-layer biases at 288 (gat_layer_fused.hip, FusedLds::HEADW)
  const size_t o_htab = reserve(296);
  const int n_rows1 = d->num_classes + nh - 1;
  const bool htab_ok = HT <= 96 && hh == 32 && n_rows1 <= 6;
  if (htab_ok) {
    std::copy(pk.begin() + o_hb0, pk.begin() + o_hb0 + HT, pk.begin() + o_htab);
    std::copy(pk.begin() + o_hW1, pk.begin() + o_hW1 + (size_t)n_rows1 * hh, pk.begin() + o_htab + 96);
    std::copy(pk.begin() + o_hb1, pk.begin() + o_hb1 + n_rows1, pk.begin() + o_htab + 288);
  }
  if ((size_t)(p - w) != n_weights) {
    set_error("internal: weight unpack consumed %zu of %zu floats", (size_t)(p - w), n_weights);
    return BGNN_ERR_INVALID;
  }

  // LocalFeatureExtractor ends in a Linear without activation (gnn.py:52-68) and GATConv's lin follows directly:
  // y = z W1^T + b1, xw = y W0^T  ==>  xw = z (W1^T W0^T) + b1 W0^T.  Folded in float64, one GEMM less per forward.
  const int HC0 = (L > 1 ? d->heads : 1) * hid;
  size_t o_l0f_Wt = reserve((size_t)hid * HC0), o_l0f_b = reserve(HC0);
  for (int o = 0; o < HC0 && gat; ++o) {
    for (int i = 0; i < hid; ++i) {
      double s = 0.0;
      for (int k = 0; k < hid; ++k) s += (double)pk[o_fe_W1t + (size_t)i * hid + k] * (double)pk[lo[0].Wt + (size_t)k * HC0 + o];
      pk[o_l0f_Wt + (size_t)i * HC0 + o] = (float)s;
    }
    double s = 0.0;
    for (int k = 0; k < hid; ++k) s += (double)pk[o_fe_b1 + k] * (double)pk[lo[0].Wt + (size_t)k * HC0 + o];
    pk[o_l0f_b + o] = (float)s;
  }

  // bf16 and float16 hi / lo images of the fused kernels' next-stage weights (layers 1.., the heads' first layers) and
  // of the folded layer-0 weight
  std::vector<size_t> o_wsp(L, 0), o_wsp16(L, 0), o_wbf(L, 0), o_wfp(L, 0);
  size_t o_hW0sp = 0, o_l0fsp = 0, o_hW0sp16 = 0, o_l0fsp16 = 0, o_hW0bf = 0, o_l0fbf = 0, o_hW0fp = 0, o_l0fpm = 0;
  bool f16_ok = true;                  // every weight fits float16: else BGNN_SPLIT_F16 falls back to the bf16 split
  std::vector<float> inv16(L, 1.0f);   // 2^-S of each float16 image (pack_split)
  float inv16_hd = 1.0f, inv16_l0f = 1.0f;
  if (gat) {
    for (int l = 1; l < L; ++l) {
      const int H = l == L - 1 ? 1 : d->heads, D = hid * d->heads, HC = H * hid;
      o_wsp[l] = reserve((size_t)D * HC); o_wsp16[l] = reserve((size_t)D * HC); o_wbf[l] = reserve((size_t)D * HC / 2);
      o_wfp[l] = reserve((size_t)D * HC);
    }
    o_hW0fp = reserve((size_t)hid * HT);
    o_hW0sp = reserve((size_t)hid * HT); o_hW0sp16 = reserve((size_t)hid * HT); o_hW0bf = reserve((size_t)hid * HT / 2);
    if (HC0 % 64 == 0) o_l0fpm = reserve((size_t)hid * HC0);
    o_l0fsp = reserve((size_t)hid * HC0); o_l0fsp16 = reserve((size_t)hid * HC0); o_l0fbf = reserve((size_t)hid * HC0 / 2 + (size_t)hid / 16 * 256 + 8);   // + the alpha tile and its constants
    for (int l = 1; l < L; ++l) {          // (reserve may reallocate pk: take the source pointers afterwards)
      const int H = l == L - 1 ? 1 : d->heads, D = hid * d->heads, HC = H * hid;
      std::vector<float> src(pk.begin() + lo[l].Wt, pk.begin() + lo[l].Wt + (size_t)D * HC);
      pack_split(src.data(), D, HC, pk.data() + o_wsp[l], false);
      if (!pack_split(src.data(), D, HC, pk.data() + o_wsp16[l], true, &inv16[l])) f16_ok = false;
      pack_bf16_image_accop(src.data(), D, HC, pk.data() + o_wbf[l]);
      pack_tilegroup_image(src.data(), D, HC, pk.data() + o_wfp[l]);
    }
    std::vector<float> src(pk.begin() + o_hW0t, pk.begin() + o_hW0t + (size_t)hid * HT);
    pack_split(src.data(), hid, HT, pk.data() + o_hW0sp, false);
    if (!pack_split(src.data(), hid, HT, pk.data() + o_hW0sp16, true, &inv16_hd)) f16_ok = false;
    pack_bf16_image_accop(src.data(), hid, HT, pk.data() + o_hW0bf);
    pack_tilegroup_image(src.data(), hid, HT, pk.data() + o_hW0fp);
    std::vector<float> src0(pk.begin() + o_l0f_Wt, pk.begin() + o_l0f_Wt + (size_t)hid * HC0);
    pack_split(src0.data(), hid, HC0, pk.data() + o_l0fsp, false);
    if (!pack_split(src0.data(), hid, HC0, pk.data() + o_l0fsp16, true, &inv16_l0f)) f16_ok = false;
    pack_bf16_image_accop(src0.data(), hid, HC0, pk.data() + o_l0fbf);
    if (o_l0fpm) pack_tilegroup_image(src0.data(), hid, HC0, pk.data() + o_l0fpm, 2);
    if (HC0 / hid <= 4) {
      const std::vector<float> b0(pk.begin() + o_l0f_b, pk.begin() + o_l0f_b + HC0);
      const std::vector<float> as0(pk.begin() + lo[0].as, pk.begin() + lo[0].as + HC0), ad0(pk.begin() + lo[0].ad, pk.begin() + lo[0].ad + HC0);
      pack_alpha_tile(src0.data(), b0.data(), as0.data(), ad0.data(), hid, HC0 / hid, hid, pk.data() + o_l0fbf + (size_t)hid * HC0 / 2);
    }
  }

  // layer 0 "aggregate first" (bf16 path, default shape): the folded lin_0 weight as four per-head [64 k][64 columns] bf16 images
  // ([head][k-step][tile] KiB, accumulator-operand k order), and layer 0's folded shift with the folded lin_0 bias carried through the
  // BatchNorm scale (the attention coefficients of a node sum to 1: sum_j alpha_ij (W h_j + b) = W sum_j alpha_ij h_j + b)
  size_t o_l0af_W = 0, o_l0af_sh = 0;
  if (gat && hid == 64 && L > 1 && d->heads == 4) {
    o_l0af_W = reserve((size_t)4 * 2048); o_l0af_sh = reserve(HC0);
    for (int hd = 0; hd < 4; ++hd) {
      std::vector<float> wh((size_t)hid * 64);
      for (int k = 0; k < hid; ++k)
        for (int c = 0; c < 64; ++c) wh[(size_t)k * 64 + c] = pk[o_l0f_Wt + (size_t)k * HC0 + hd * 64 + c];
      pack_bf16_image_accop(wh.data(), hid, 64, pk.data() + o_l0af_W + (size_t)hd * 2048);
    }
    for (int o = 0; o < HC0; ++o)
      pk[o_l0af_sh + o] = (float)((double)pk[lo[0].sh + o] + (double)pk[lo[0].sc + o] * (double)pk[o_l0f_b + o]);
  }

  // plain backbones (hidden 64): the layer weight in the fused layer kernel's column-permuted image (launch_fused_plain_layer)
  std::vector<size_t> o_plainfp(L, 0);
  if (!gat && hid == 64) {
    for (int l = 0; l < L; ++l) o_plainfp[l] = reserve((size_t)(d->gnn_type == BGNN_GNN_SAGE ? 2 : 1) * hid * hid);
    for (int l = 0; l < L; ++l) {                          // (reserve may reallocate pk: sources taken afterwards)
      const int D = (d->gnn_type == BGNN_GNN_SAGE ? 2 : 1) * hid;
      std::vector<float> src(pk.begin() + lo[l].Wt, pk.begin() + lo[l].Wt + (size_t)D * hid);
      pack_tilegroup_image(src.data(), D, hid, pk.data() + o_plainfp[l]);
    }
  }

  // layers wider than 256 columns: blocked images for the generic GEMM (layer 0: the folded and the unfolded weight)
  std::vector<size_t> o_wblk(L, 0);
  size_t o_l0f_blk = 0;
  if (gat && d->heads * hid > 256) {
    for (int l = 0; l + 1 < L; ++l) o_wblk[l] = reserve((size_t)(l == 0 ? hid : hid * d->heads) * d->heads * hid);
    if (L > 1) o_l0f_blk = reserve((size_t)hid * HC0);
    for (int l = 0; l + 1 < L; ++l) {                      // (reserve may reallocate pk: sources taken afterwards)
      const int D = l == 0 ? hid : hid * d->heads, HC = d->heads * hid;
      std::vector<float> src(pk.begin() + lo[l].Wt, pk.begin() + lo[l].Wt + (size_t)D * HC);
      pack_col_blocks(src.data(), D, HC, pk.data() + o_wblk[l]);
    }
    if (o_l0f_blk) {
      std::vector<float> src0(pk.begin() + o_l0f_Wt, pk.begin() + o_l0f_Wt + (size_t)hid * HC0);
      pack_col_blocks(src0.data(), hid, HC0, pk.data() + o_l0f_blk);
    }
  }

  bgnn_model *m = new bgnn_model();
  m->ctx = ctx; m->desc = *d; m->blob_floats = pk.size();
  hipError_t e = hipMalloc((void **)&m->blob, pk.size() * sizeof(float));
  if (e != hipSuccess) { delete m; set_error("hipMalloc(model) failed: %s", hipGetErrorString(e)); return BGNN_ERR_NOMEM; }
  e = hipMemcpy(m->blob, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(m->blob); delete m; set_error("hipMemcpy(model) failed: %s", hipGetErrorString(e)); return BGNN_ERR_HIP; }
  m->fe_W0t = m->blob + o_fe_W0t; m->fe_b0 = m->blob + o_fe_b0; m->fe_W1t = m->blob + o_fe_W1t; m->fe_b1 = m->blob + o_fe_b1;
  m->l0f_Wt = m->blob + o_l0f_Wt; m->l0f_b = m->blob + o_l0f_b;
  m->l0f_Wsp = gat ? m->blob + o_l0fsp : nullptr;
  m->l0f_Wsp16 = gat && f16_ok ? m->blob + o_l0fsp16 : nullptr;
  m->l0f_Wsp16_inv = inv16_l0f; m->hd_W0sp16_inv = inv16_hd;
  m->l0f_Wbf = gat ? m->blob + o_l0fbf : nullptr;
  m->l0f_Wpm = gat && o_l0fpm ? m->blob + o_l0fpm : nullptr;
  m->l0f_Wt_blk = o_l0f_blk ? m->blob + o_l0f_blk : nullptr;
  m->hd_W0bf = gat ? m->blob + o_hW0bf : nullptr;
  m->hd_W0fp = gat ? m->blob + o_hW0fp : nullptr;
  m->l0af_W = o_l0af_W ? m->blob + o_l0af_W : nullptr;
  m->l0af_shift = o_l0af_sh ? m->blob + o_l0af_sh : nullptr;
  m->layers.resize(L);
  for (int l = 0; l < L && !gat; ++l) {
    BgnnLayer &Ly = m->layers[l];
    Ly = BgnnLayer{};
    Ly.heads = 1; Ly.d_in = hid; Ly.width = hid; Ly.concat = l != L - 1;
    Ly.Wt = m->blob + lo[l].Wt;
    if (d->gnn_type == BGNN_GNN_GCN) { Ly.scale = m->blob + lo[l].sc; Ly.shift = m->blob + lo[l].sh; }
    if (d->gnn_type == BGNN_GNN_SAGE) Ly.b2 = m->blob + lo[l].b2;
    if (d->gnn_type == BGNN_GNN_GIN) { Ly.b1 = m->blob + lo[l].b1; Ly.Wt2 = m->blob + lo[l].Wt2; Ly.b2 = m->blob + lo[l].b2; }
    Ly.tr_bias = m->blob + lo[l].tr_bias; Ly.bn_w = m->blob + lo[l].tr_bw; Ly.bn_b = m->blob + lo[l].tr_bb;
    Ly.tr_Wt = lo[l].tr_Wt ? m->blob + lo[l].tr_Wt : nullptr;
    Ly.Wfp = o_plainfp[l] ? m->blob + o_plainfp[l] : nullptr;
  }
  m->ones = m->blob + o_ones;
  for (int l = 0; l < L && gat; ++l) {
    const bool last = l == L - 1;
    BgnnLayer &Ly = m->layers[l];
    Ly.heads = last ? 1 : d->heads; Ly.d_in = l == 0 ? hid : hid * d->heads;
    Ly.width = last ? hid : Ly.heads * hid; Ly.concat = !last;
    Ly.Wt = m->blob + lo[l].Wt; Ly.att_src = m->blob + lo[l].as; Ly.att_dst = m->blob + lo[l].ad;
    Ly.Wt_blk = o_wblk[l] ? m->blob + o_wblk[l] : nullptr;
    Ly.V = m->blob + lo[l].V; Ly.scale = m->blob + lo[l].sc; Ly.shift = m->blob + lo[l].sh;
    Ly.Wsp = l > 0 ? m->blob + o_wsp[l] : nullptr;
    Ly.Wsp16 = l > 0 && f16_ok ? m->blob + o_wsp16[l] : nullptr;
    Ly.Wsp16_inv = inv16[l];
    Ly.Wbf = l > 0 ? m->blob + o_wbf[l] : nullptr;
    Ly.Wfp = l > 0 ? m->blob + o_wfp[l] : nullptr;
    Ly.tr_bias = m->blob + lo[l].tr_bias; Ly.bn_w = m->blob + lo[l].tr_bw; Ly.bn_b = m->blob + lo[l].tr_bb;
  }
  for (int l = 0; l < L && gat; ++l) {                 // host copy of the folded edge vectors (model_canonical_V)
    const int H = l == L - 1 ? 1 : d->heads;
    m->h_V.insert(m->h_V.end(), pk.begin() + lo[l].V, pk.begin() + lo[l].V + (size_t)H * ED);
  }
  m->head_hidden_total = HT;
  m->hd_W0sp = gat ? m->blob + o_hW0sp : nullptr;
  m->hd_W0sp16 = gat && f16_ok ? m->blob + o_hW0sp16 : nullptr;
  m->hd_W0t = m->blob + o_hW0t; m->hd_b0 = m->blob + o_hb0; m->hd_W1 = m->blob + o_hW1; m->hd_b1 = m->blob + o_hb1;
  m->hd_tab = htab_ok ? m->blob + o_htab : nullptr;
  *out = m;
  return BGNN_OK;
}

int bgnn_model_destroy(bgnn_model *m) {
  if (!m) return BGNN_OK;
  (void)hipSetDevice(m->ctx->device);
  (void)hipStreamSynchronize(m->ctx->stream);
  (void)hipFree(m->blob);
  for (auto &kv : m->v3_tables) (void)hipFree(kv.second);
  delete m;
  return BGNN_OK;
}

// The edge vectors of every GAT layer over the canonical attributes (distance, depth_difference, slope) for a graph built with
// another edge feature list: device table [layers][heads][3], V3[l][h][id] = sum over the list positions j with ids[j] == id of
// V_l[h][j] ("zero" columns drop out, a repeated attribute adds up).  Made once per (model, list) and kept with the model.
static int model_canonical_V(bgnn_model *m, const bgnn_graph *g, const float **out) {
  const int ED = g->ED, heads = m->desc.heads;
  uint32_t key = (uint32_t)ED;
  for (int j = 0; j < 4; ++j) key = key * 8u + (uint32_t)(j < ED ? g->edge_ids[j] : 7);
  for (auto &kv : m->v3_tables)
    if (kv.first == key) { *out = kv.second; return BGNN_OK; }
  const size_t L = m->layers.size();
  std::vector<float> t(L * (size_t)heads * 3, 0.0f);
  size_t off = 0;
  for (size_t l = 0; l < L; ++l) {
    const int H = m->layers[l].heads;
    for (int h = 0; h < H; ++h)
      for (int j = 0; j < ED; ++j) {
        const int id = g->edge_ids[j];
        if (id >= 0 && id < 3) t[(l * heads + h) * 3 + id] += m->h_V[off + (size_t)h * ED + j];
      }
    off += (size_t)H * ED;
  }
  float *d = nullptr;
  BGNN_HIP_CHECK(hipMalloc((void **)&d, t.size() * sizeof(float)));
  if (hipMemcpy(d, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(d);
    set_error("hipMemcpy(canonical edge vectors) failed");
    return BGNN_ERR_HIP;
  }
  m->v3_tables.emplace_back(key, d);
  *out = d;
  return BGNN_OK;
}

// ---- graph ------------------------------------------------------------------------------------
static void graph_free(bgnn_graph *g) {
  DevPool &P = g->ctx->pool;
  g->ctx->live_graphs.erase(g);
  if (g->d_tables) P.release(g->d_tables);       // ragged batch: tiles / items / items2 / items3 / canvas tables live in this block
  else if (!g->tables_cached) { P.release(g->d_tiles); P.release(g->d_items); }
  else
    for (auto &e : g->ctx->table_cache)          // drop this graph's reference on its cache entry (evictable at 0)
      if (e.id == g->table_cache_id) { if (e.refs > 0) --e.refs; break; }
  P.release(g->d_node_id); P.release(g->d_cell_of_node);
  P.release(g->d_counts); P.release(g->d_x8); P.release(g->d_local_std); P.release(g->d_nbr);
  P.release(g->d_eattr); P.release(g->d_rowptr); P.release(g->d_edge_perm);
  P.release(g->d_slope); P.release(g->d_node_depth); P.release(g->d_tile_dist); P.release(g->d_atlas_tile_of);
  P.release(g->d_atlas);
  delete g;
}

static int validate_opts(const bgnn_graph_opts *o) {
  BGNN_REQUIRE(o->connectivity == 4 || o->connectivity == 8 || o->connectivity == 16,
               "Unknown connectivity: %d", o->connectivity);
  BGNN_REQUIRE(o->n_node_features >= 0 && o->n_node_features <= 8, "n_node_features=%d out of range", o->n_node_features);
  BGNN_REQUIRE(o->n_edge_features >= 1 && o->n_edge_features <= 4, "n_edge_features=%d out of range (1..4)", o->n_edge_features);
  for (int i = 0; i < o->n_node_features; ++i)
    BGNN_REQUIRE(o->node_features[i] >= 0 && o->node_features[i] <= 7, "bad node feature id %d", o->node_features[i]);
  for (int i = 0; i < o->n_edge_features; ++i)
    BGNN_REQUIRE(o->edge_features[i] >= 0 && o->edge_features[i] <= 3, "bad edge feature id %d", o->edge_features[i]);
  return BGNN_OK;
}

static int graph_build_impl(bgnn_ctx *ctx, const bgnn_tiles *tiles, const bgnn_graph_opts *opts, bgnn_graph **out, int64_t *n_nodes_copy,
                            float *const *clear_grids = nullptr);

int bgnn_graph_build(bgnn_ctx *ctx, const bgnn_tiles *tiles, const bgnn_graph_opts *opts, bgnn_graph **out) {
  return graph_build_impl(ctx, tiles, opts, out, nullptr);
}

static int graph_build_impl(bgnn_ctx *ctx, const bgnn_tiles *tiles, const bgnn_graph_opts *opts, bgnn_graph **out, int64_t *n_nodes_copy,
                            float *const *clear_grids) {
  BGNN_REQUIRE(ctx && tiles && opts && out, "bgnn_graph_build: NULL argument");
  BGNN_REQUIRE(tiles->n_tiles >= 1, "bgnn_graph_build: n_tiles=%d", tiles->n_tiles);
  // (the per-grid kernels index grids by gridDim.y: 65 535 at most -- say so instead of failing the launch)
  BGNN_REQUIRE(tiles->n_tiles <= 60000, "bgnn_graph_build: %d grids in one batch, at most 60000 (split the batch)", tiles->n_tiles);
  BGNN_REQUIRE(tiles->hw && tiles->resolution && tiles->depth && tiles->mask, "bgnn_graph_build: NULL tile array");
  BGNN_TRY(validate_opts(opts));
  BGNN_HIP_CHECK(hipSetDevice(ctx->device));
  bgnn_graph *g = new bgnn_graph();
  g->ctx = ctx; g->kind = 0; g->n_tiles = tiles->n_tiles; g->d_n_nodes_copy = n_nodes_copy;
  if (clear_grids) for (int i = 0; i < 3; ++i) g->clear_grids[i] = clear_grids[i];
  g->K = opts->connectivity; g->ED = opts->n_edge_features; g->include_self_loops = opts->include_self_loops ? 1 : 0;
  g->has_unc = tiles->uncertainty ? 1 : 0;
  // feature count (see launch_graph_build for the column rule)
  {
    int nf = 0; bool listed = false;
    for (int i = 0; i < opts->n_node_features; ++i) {
      if (opts->node_features[i] == BGNN_NF_UNCERTAINTY) { listed = true; if (!tiles->uncertainty) continue; }
      ++nf;
    }
    if (tiles->uncertainty && !listed) ++nf;
    if (nf > 8 || nf < 1) { delete g; set_error("node feature count %d out of range (1..8)", nf); return BGNN_ERR_INVALID; }
    g->F = nf;
  }
  int64_t cells = 0;
  int64_t cells_est = 0;
  for (int t = 0; t < tiles->n_tiles; ++t) cells_est += (int64_t)tiles->hw[2 * t] * tiles->hw[2 * t + 1];
  const int item_cells = (int)std::min<int64_t>(2048, std::max<int64_t>(256, cells_est / (4 * ctx->num_cus)));
  std::vector<BgnnWorkItem> items, items2, items3;
  bool uniform = true;
  g->h_tiles.resize(tiles->n_tiles);
  for (int t = 0; t < tiles->n_tiles; ++t) {
    const int h = tiles->hw[2 * t], w = tiles->hw[2 * t + 1];
    if (h < 2 || w < 2) {
      // np.gradient needs >= 2 samples per axis: the reference raises ValueError here too
      delete g;
      set_error("Shape of array too small to calculate a numerical gradient, at least (edge_order + 1) elements are "
                "required. (tile %d is %dx%d)", t, h, w);
      return BGNN_ERR_INVALID;
    }
    BgnnTileMeta &m = g->h_tiles[t];
    m.h = h; m.w = w; m.cell_off = (int32_t)cells; m.pad = 0;
    m.rx = tiles->resolution[2 * t]; m.ry = tiles->resolution[2 * t + 1];
    cells += (int64_t)h * w;
    if (cells >= ((int64_t)1 << 30)) { delete g; set_error("batch too large: 2^30 cells or more; split it"); return BGNN_ERR_INVALID; }
    // row bands of the feature kernel: ~2048 cells each for big batches, down to one 256-thread pass (256 cells) when the
    // batch is small, so that a single tile still spreads over the whole chip
    int rows_per = std::max(1, item_cells / w);
    for (int r0 = 0; r0 < h; r0 += rows_per) items.push_back({t, r0, std::min(rows_per, h - r0), 0});
    if (h != tiles->hw[0] || w != tiles->hw[1]) uniform = false;
    if (w > g->max_w) g->max_w = w;
  }
  if (uniform) {
    g->uni_h = tiles->hw[0]; g->uni_w = tiles->hw[1];
    g->bh2 = (g->uni_h + 15) / 16; g->bw2 = (g->uni_w + 15) / 16;
    g->n_blocks2 = g->n_tiles * g->bh2 * g->bw2;
    g->bh3 = (g->uni_h + 7) / 8; g->bw3 = g->bw2;
    g->n_blocks3 = g->n_tiles * g->bh3 * g->bw3;
  } else {
    for (int t = 0; t < tiles->n_tiles; ++t)
      for (int r0 = 0; r0 < g->h_tiles[t].h; r0 += 16)
        for (int c0 = 0; c0 < g->h_tiles[t].w; c0 += 16) items2.push_back({t, r0, c0, 0});
    g->n_blocks2 = (int32_t)items2.size();
    for (int t = 0; t < tiles->n_tiles; ++t)
      for (int r0 = 0; r0 < g->h_tiles[t].h; r0 += 8)
        for (int c0 = 0; c0 < g->h_tiles[t].w; c0 += 16) items3.push_back({t, r0, c0, 0});
    g->n_blocks3 = (int32_t)items3.size();
  }
  // ragged batch: shelf-pack the grids (tallest first) onto a canvas for the fused layer kernels
  std::vector<int32_t> atlas_pos;
  BgnnTileMeta atlas_meta{};
  if (!uniform && tiles->n_tiles >= 2 && ctx->opts.ragged_atlas) {
    const int G = g->K == 16 ? 2 : 1;                                   // gutter = reach of the stencil
    // First-fit shelves, grids by decreasing height (wider first among equals): a grid goes into the first shelf that is tall enough
    // and has room, else it opens a new shelf.  A few canvas widths around sqrt(cells) are tried and the one with the fewest 8 x 16
    // blocks wins -- a 50 000-node batch of refinement grids then needs ~480 blocks instead of ~525 (one width, next-fit), i.e. ONE
    // round of workgroups on 512 slots instead of one and a bit.  (~10 us of host work per batch.)
    std::vector<int> order(tiles->n_tiles);
    for (int t = 0; t < tiles->n_tiles; ++t) order[t] = t;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
      return g->h_tiles[a].h != g->h_tiles[b].h ? g->h_tiles[a].h > g->h_tiles[b].h : g->h_tiles[a].w > g->h_tiles[b].w;
    });
    struct Shelf { int y, h, x; };
    std::vector<Shelf> shelves;
    auto pack = [&](int aw, int32_t *pos) -> int {                      // -> canvas height (before rounding to whole blocks)
      shelves.clear();
      int y = 0;
      for (int t : order) {
        const int h = g->h_tiles[t].h, w = g->h_tiles[t].w;
        bool placed = false;
        for (auto &sh : shelves)
          if (sh.x + w <= aw && h <= sh.h) {
            if (pos) { pos[2 * t] = sh.y; pos[2 * t + 1] = sh.x; }
            sh.x += w + G; placed = true;
            break;
          }
        if (!placed) {
          if (pos) { pos[2 * t] = y; pos[2 * t + 1] = 0; }
          shelves.push_back({y, h, w + G});
          y += h + G;
        }
      }
      return y - G;
    };
    const int aw_min = ((g->max_w + G + 15) / 16) * 16;
    const int side = (int)std::sqrt((double)cells);
    int aw = 0, ah = 0;
    int64_t best = -1;
    // <= 9 candidates between 0.9 and 1.5 sqrt(cells) (<= 4 for batches of many hundred grids: there the host time counts and a round more or less does not)
    const int step = std::max(16, (side * 6 / 10 / (tiles->n_tiles > 400 ? 3 : 8) + 15) / 16 * 16);
    for (int cand = std::max(aw_min, (side * 9 / 10 + 15) / 16 * 16);; cand += step) {
      const int hc = ((pack(cand, nullptr) + 7) / 8) * 8;
      const int64_t blocks = (int64_t)(hc / 8) * (cand / 16);
      if (best < 0 || blocks < best) { best = blocks; aw = cand; ah = hc; }
      if (cand >= side * 3 / 2 || cand + step > 8192) break;            // (always at least one candidate)
    }
    atlas_pos.assign((size_t)tiles->n_tiles * 2, 0);
    (void)pack(aw, atlas_pos.data());
    // worth it only if the canvas has fewer blocks than the grids have on their own (refinement grids: yes; two big tiles of
    // different size: no -- they fill their own blocks already and would leave half a canvas empty)
    if ((int64_t)(ah / 8) * (aw / 16) < (int64_t)items3.size()) {
      g->atlas_w = aw; g->atlas_h = ah;
      atlas_meta.h = g->atlas_h; atlas_meta.w = g->atlas_w; atlas_meta.cell_off = 0; atlas_meta.rx = 1.0; atlas_meta.ry = 1.0;
    }
  }
  g->total_cells = (int32_t)cells; g->row_capacity = (int32_t)cells; g->n_items = (int32_t)items.size();
  DevPool &P = ctx->pool;
  int rc = BGNN_OK;
#define GALLOC(ptr, type, count) if (rc == BGNN_OK) { void *_p = nullptr; rc = P.alloc((size_t)(count) * sizeof(type), &_p); ptr = (type *)_p; }
  // uniform batches at one resolution: the two tables come from (or go into) the context's cache
  bgnn_ctx::TableCache *hit = nullptr;
  bool cacheable = uniform;
  for (int t = 1; t < tiles->n_tiles && cacheable; ++t)
    cacheable = tiles->resolution[2 * t] == tiles->resolution[0] && tiles->resolution[2 * t + 1] == tiles->resolution[1];
  if (cacheable)
    for (auto &e : ctx->table_cache)
      if (e.n_tiles == g->n_tiles && e.h == g->uni_h && e.w == g->uni_w && e.item_cells == item_cells &&
          e.rx == tiles->resolution[0] && e.ry == tiles->resolution[1]) { hit = &e; break; }
  if (hit) {
    g->d_tiles = hit->d_tiles; g->d_items = hit->d_items; g->tables_cached = true; g->table_cache_id = hit->id;
    ++hit->refs;
    hit->stamp = ++ctx->table_stamp;
  } else if (cacheable) {
    if (ctx->table_cache.size() >= 8) {                    // evict the least recently used entry NO LIVE GRAPH points into
      size_t lru = ctx->table_cache.size();                // (nothing in flight reads it after a stream sync)
      for (size_t i = 0; i < ctx->table_cache.size(); ++i)
        if (ctx->table_cache[i].refs == 0 && (lru == ctx->table_cache.size() || ctx->table_cache[i].stamp < ctx->table_cache[lru].stamp)) lru = i;
      if (lru == ctx->table_cache.size()) cacheable = false;          // every entry is pinned: this graph keeps private tables
      else {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(ctx->table_cache[lru].d_tiles); (void)hipFree(ctx->table_cache[lru].d_items);
        ctx->table_cache.erase(ctx->table_cache.begin() + lru);
      }
    }
    if (cacheable) {
      bgnn_ctx::TableCache e{g->n_tiles, g->uni_h, g->uni_w, item_cells, tiles->resolution[0], tiles->resolution[1], nullptr, nullptr,
                             g->n_items, ++ctx->table_stamp, ctx->table_next_id++, 1};
      if (hipMalloc((void **)&e.d_tiles, sizeof(BgnnTileMeta) * g->n_tiles) == hipSuccess &&
          hipMalloc((void **)&e.d_items, sizeof(BgnnWorkItem) * g->n_items) == hipSuccess) {
        ctx->table_cache.push_back(e);
        g->d_tiles = e.d_tiles; g->d_items = e.d_items; g->tables_cached = true; g->table_cache_id = e.id;
      } else {
        (void)hipGetLastError();
        if (e.d_tiles) (void)hipFree(e.d_tiles);
        cacheable = false;
      }
    }
  }
  // ragged batches: every host-built table travels in ONE block (one staging copy, one H2D) -- with 50 000-node VR batches
  // the per-batch host work is what bounds the stream
  std::vector<char> blockh;
  size_t o_tiles = 0, o_items = 0, o_items2 = 0, o_items3 = 0, o_atile = 0, o_apos = 0;
  if (!uniform) {
    auto put = [&](const void *p, size_t bytes) {
      const size_t at = (blockh.size() + 255) & ~(size_t)255;
      blockh.resize(at + bytes);
      memcpy(blockh.data() + at, p, bytes);
      return at;
    };
    o_tiles = put(g->h_tiles.data(), sizeof(BgnnTileMeta) * g->n_tiles);
    o_items = put(items.data(), sizeof(BgnnWorkItem) * items.size());
    o_items2 = put(items2.data(), sizeof(BgnnWorkItem) * items2.size());
    if (g->atlas_h) {
      o_atile = put(&atlas_meta, sizeof(atlas_meta));
      o_apos = put(atlas_pos.data(), sizeof(int32_t) * atlas_pos.size());
    } else {
      o_items3 = put(items3.data(), sizeof(BgnnWorkItem) * items3.size());
    }
    GALLOC(g->d_tables, char, blockh.size())
    if (rc == BGNN_OK) {
      g->d_tiles = (BgnnTileMeta *)(g->d_tables + o_tiles); g->d_items = (BgnnWorkItem *)(g->d_tables + o_items);
      g->d_items2 = (BgnnWorkItem *)(g->d_tables + o_items2);
      if (g->atlas_h) { g->d_atlas_tile = (BgnnTileMeta *)(g->d_tables + o_atile); g->d_atlas_pos = (int32_t *)(g->d_tables + o_apos); }
      else g->d_items3 = (BgnnWorkItem *)(g->d_tables + o_items3);
    }
  } else if (!g->tables_cached) {
    GALLOC(g->d_tiles, BgnnTileMeta, g->n_tiles)
    GALLOC(g->d_items, BgnnWorkItem, g->n_items)
  }
  GALLOC(g->d_node_id, int32_t, cells)
  GALLOC(g->d_cell_of_node, int32_t, cells)
  GALLOC(g->d_counts, int64_t, 4)
  GALLOC(g->d_x8, float, cells * 8)
  GALLOC(g->d_local_std, float, cells)
  GALLOC(g->d_nbr, int32_t, cells * g->K)
  // default edge feature list on a stencil the fused kernels know: compact edge storage (graph_build.hip, FeatureArgs)
  // every edge feature list is a selection / ordering of (distance, depth_difference, slope, zero): all of them are built compact
  // (round 3: the default list only -- any other list carried the full table and ran on the unfused kernels)
  g->compact_edges = g->K == 4 || g->K == 8 || g->K == 16;
  for (int i = 0; i < 4; ++i) g->edge_ids[i] = i < g->ED ? opts->edge_features[i] : BGNN_EF_ZERO;
  g->edge_default = g->ED == 3 && opts->edge_features[0] == BGNN_EF_DISTANCE && opts->edge_features[1] == BGNN_EF_DEPTH_DIFFERENCE &&
                    opts->edge_features[2] == BGNN_EF_SLOPE;
  if (g->compact_edges) {
    GALLOC(g->d_slope, float, cells * g->K)
    GALLOC(g->d_node_depth, float, cells)
    GALLOC(g->d_tile_dist, float4, g->n_tiles)
  } else {
    GALLOC(g->d_eattr, float, cells * g->K * g->ED)
  }
  if (g->atlas_h) GALLOC(g->d_atlas, int32_t, (size_t)g->atlas_h * g->atlas_w)
  if (g->atlas_h && g->compact_edges) GALLOC(g->d_atlas_tile_of, int32_t, (size_t)g->atlas_h * g->atlas_w)
#undef GALLOC
  if (rc == BGNN_OK && !uniform) rc = ctx_upload(ctx, blockh.data(), blockh.size(), g->d_tables);
  if (rc == BGNN_OK && uniform && !hit) rc = ctx_upload(ctx, g->h_tiles.data(), sizeof(BgnnTileMeta) * g->n_tiles, g->d_tiles);
  if (rc == BGNN_OK && uniform && !hit) rc = ctx_upload(ctx, items.data(), sizeof(BgnnWorkItem) * items.size(), g->d_items);
  if (rc == BGNN_OK) rc = launch_graph_build(ctx, g, tiles, opts);
  if (rc != BGNN_OK) { graph_free(g); return rc; }
  ctx->live_graphs.insert(g);
  *out = g;
  return BGNN_OK;
}

int bgnn_graph_from_edges(bgnn_ctx *ctx, int64_t n_nodes, int32_t n_feat, const float *x, int64_t n_edges,
                          const int64_t *edge_index, int32_t edge_dim, const float *edge_attr, bgnn_graph **out) {
  BGNN_REQUIRE(ctx && out, "bgnn_graph_from_edges: NULL argument");
  BGNN_REQUIRE(n_nodes >= 0 && n_nodes < ((int64_t)1 << 31) && n_edges >= 0 && n_edges < ((int64_t)1 << 31),
               "graph too large");
  BGNN_REQUIRE(n_feat >= 1 && n_feat <= 8, "n_feat=%d unsupported (1..8)", n_feat);
  BGNN_REQUIRE(edge_dim >= 1 && edge_dim <= 4, "edge_dim=%d unsupported (1..4)", edge_dim);
  BGNN_REQUIRE((x || n_nodes == 0) && (edge_index || n_edges == 0) && (edge_attr || n_edges == 0), "NULL tensor");
  BGNN_HIP_CHECK(hipSetDevice(ctx->device));
  bgnn_graph *g = new bgnn_graph();
  g->ctx = ctx; g->kind = 1; g->F = n_feat; g->ED = edge_dim; g->K = 0;
  g->total_cells = (int32_t)n_nodes; g->row_capacity = (int32_t)n_nodes; g->generic_E = n_edges;
  int rc = launch_generic_build(ctx, g, n_nodes, n_feat, x, n_edges, edge_index, edge_dim, edge_attr);
  if (rc != BGNN_OK) { graph_free(g); return rc; }
  ctx->live_graphs.insert(g);
  *out = g;
  return BGNN_OK;
}

int bgnn_graph_destroy(bgnn_graph *g) {
  if (!g) return BGNN_OK;
  graph_free(g);
  return BGNN_OK;
}

int bgnn_graph_counts(bgnn_graph *g, int64_t *n_nodes, int64_t *n_edges, int32_t *n_feat, int32_t *edge_dim,
                      int64_t *node_off, int64_t *edge_off) {
  BGNN_REQUIRE(g, "graph is NULL");
  bgnn_ctx *ctx = g->ctx;
  BGNN_HIP_CHECK(hipSetDevice(ctx->device));
  if (n_feat) *n_feat = g->F;
  if (edge_dim) *edge_dim = g->ED;
  if (g->kind == 1) {
    if (n_nodes) *n_nodes = g->total_cells;
    if (n_edges) *n_edges = g->generic_E;
    if (node_off) { node_off[0] = 0; node_off[1] = g->total_cells; }
    if (edge_off) { edge_off[0] = 0; edge_off[1] = g->generic_E; }
    return BGNN_OK;
  }
  BGNN_TRY(launch_graph_count_edges(g));
  std::vector<int64_t> tc((size_t)g->n_tiles * 2);
  BGNN_HIP_CHECK(hipMemcpyAsync(tc.data(), ctx->ws[5], tc.size() * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  BGNN_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  int64_t nn = 0, ne = 0;
  for (int t = 0; t < g->n_tiles; ++t) {
    if (node_off) node_off[t] = nn;
    if (edge_off) edge_off[t] = ne;
    nn += tc[t]; ne += tc[g->n_tiles + t];
  }
  if (node_off) node_off[g->n_tiles] = nn;
  if (edge_off) edge_off[g->n_tiles] = ne;
  g->n_nodes_host = nn; g->n_edges_host = ne;
  if (n_nodes) *n_nodes = nn;
  if (n_edges) *n_edges = ne;
  return BGNN_OK;
}

int bgnn_graph_export(bgnn_graph *g, float *x, int64_t *edge_index, float *edge_attr, float *pos,
                      int64_t *valid_rows, int64_t *valid_cols, float *local_std, int64_t *batch) {
  BGNN_REQUIRE(g, "graph is NULL");
  BGNN_REQUIRE(g->kind == 0, "bgnn_graph_export: only graphs built by bgnn_graph_build can be exported");
  BGNN_HIP_CHECK(hipSetDevice(g->ctx->device));
  return launch_graph_export(g, x, edge_index, edge_attr, pos, valid_rows, valid_cols, local_std, batch);
}

int bgnn_graph_scatter(bgnn_graph *g, const float *node_values, float fill, float *grid) {
  BGNN_REQUIRE(g && node_values && grid, "bgnn_graph_scatter: NULL argument");
  BGNN_REQUIRE(g->kind == 0, "Data object missing grid_shape metadata");
  BGNN_HIP_CHECK(hipSetDevice(g->ctx->device));
  return launch_graph_scatter(g, node_values, fill, grid);
}

// ---- forward ----------------------------------------------------------------------------------
struct GridOut {             // optional fused node -> grid outputs (bgnn_infer_tiles)
  float *cls = nullptr, *conf = nullptr, *corr = nullptr;
  float norm_floor = 0.01f;
  bool done = false;         // set when the fused tail wrote the grids
};

// training-mode forward: BatchNorm statistics of this batch, written layer by layer ([sum of layer widths] each)
struct TrainOut {
  float *mean, *var_unbiased;
  const bgnn_dropout *dp = nullptr;      // active dropout (bgnn_forward_train_dropout)
};

static int forward_impl(bgnn_ctx *ctx, bgnn_model *m, bgnn_graph *g, float thr_auto, float thr_review,
                        const bgnn_outputs *o, GridOut *grids, TrainOut *tr = nullptr) {
  const bgnn_model_desc &d = m->desc;
  BGNN_REQUIRE(g->F == d.in_channels, "mat1 and mat2 shapes cannot be multiplied (graph has %d node features, model expects %d)",
               g->F, d.in_channels);
  const bool gat = d.gnn_type == BGNN_GNN_GAT;
  BGNN_REQUIRE(!gat || g->ED == d.edge_dim, "edge_attr has %d columns, model edge_dim is %d", g->ED, d.edge_dim);
  const int64_t rows = g->row_capacity;
  if (rows <= 0) return BGNN_OK;
  const int hid = d.hidden;
  const int maxw = std::max(2 * hid, d.heads * hid);
  void *pa, *pb, *pasd, *phid;
  BGNN_TRY(ctx_workspace(ctx, 0, (size_t)rows * maxw * sizeof(float), &pa));
  BGNN_TRY(ctx_workspace(ctx, 1, (size_t)rows * maxw * sizeof(float), &pb));
  BGNN_TRY(ctx_workspace(ctx, 2, (size_t)rows * 4 * d.heads * sizeof(float), &pasd));
  BGNN_TRY(ctx_workspace(ctx, 3, (size_t)rows * m->head_hidden_total * sizeof(float), &phid));
  float *X = (float *)pa, *Y = (float *)pb, *hidb = (float *)phid;
  float *asdX = (float *)pasd, *asdY = asdX + rows * 2 * d.heads;
  const int64_t *dm = g->d_counts;
  const bool use_fused = ctx->opts.fused && !tr;
  // matrix_path 3 (BASELINE config 3): layer activations xw are stored as bf16 and multiplied on the bf16 MFMA; it exists
  // only on the fused stencil path of the default model shape and only in eval mode
  const bool bf16 = ctx->opts.matrix_path == 3 && !tr;
  if (bf16) {
    BGNN_REQUIRE(gat && use_fused && g->kind == 0 && hid == 64 && d.heads == 4 && d.num_layers >= 2 && g->compact_edges && !o->hidden,
                 "matrix_path = bf16 (bf16 activation storage) runs on the fused stencil path of the default model shape only "
                 "(GAT, hidden 64, heads 4, >= 2 layers, graphs built by bgnn_graph_build)");
  }   // the fused layers carry the folded eval statistics
  void *bnws = nullptr;
  if (tr) BGNN_TRY(ctx_workspace(ctx, 5, bn_train_workspace_bytes(maxw >= 256 ? 256 : maxw), &bnws));
  size_t tr_off = 0;
  // training mode with active dropout: x [rows][width] *= keep / (1 - p) in place (stream ids: bgnn.h, bgnn_dropout)
  const bgnn_dropout *dp = tr ? tr->dp : nullptr;
  auto drop = [&](float *x, int width, float p, uint32_t stream) {
    return dp && p > 0.0f ? launch_dropout(ctx, x, width, width, dm, rows, make_drop_spec(p, dp->seed, stream)) : BGNN_OK;
  };
  auto batch_norm = [&](float *z, const BgnnLayer &L, int relu) {          // z [rows][L.width], in place (256 columns per launch)
    int rc = BGNN_OK;
    for (int c0 = 0; c0 < L.width && rc == BGNN_OK; c0 += 256) {
      const int w = std::min(256, L.width - c0);
      rc = launch_bn_train(ctx, z + c0, L.width, w, rows, dm, L.bn_w + c0, L.bn_b + c0, d.bn_eps, relu, bnws,
                           tr->mean ? tr->mean + tr_off + c0 : nullptr, tr->var_unbiased ? tr->var_unbiased + tr_off + c0 : nullptr);
    }
    tr_off += (size_t)L.width;
    return rc;
  };
  // feature extractor (gnn.py:386): Linear(in,hid) ReLU [Dropout] Linear(hid,hid); then lin of layer 0
  if (!gat && g->kind != 0 && d.gnn_type != BGNN_GNN_GCN) {
    // foreign graphs: the CSR build dropped explicit self loops (GATConv and GCNConv replace them anyway); SAGEConv and
    // GINConv treat them as ordinary edges, which the CSR no longer holds
    int64_t c[4];
    BGNN_HIP_CHECK(hipMemcpyAsync(c, g->d_counts, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
    BGNN_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (c[2] != c[1]) {
      set_error("GraphSAGE / GIN on a foreign graph with explicit self loops (or out-of-range edges: %lld of %lld edges kept) "
                "is not supported", (long long)c[2], (long long)c[1]);
      return BGNN_ERR_UNSUPPORTED;
    }
  }
  bool layer0_done = false;                           // bf16 path: layer 0's aggregate already ran (aggregate-first launch): the loop starts at layer 1
  const size_t nl_gat = gat ? m->layers.size() : 0;
  if (!gat) {
    // GCN / GraphSAGE / GIN backbones (gnn.py:120-143; torch_geometric default arguments): plain gathers + GEMMs.
    // Not the hot path: no fusion beyond BatchNorm / bias / ReLU folded into the neighbouring kernel.
    BGNN_TRY(launch_gemm_f32(ctx, g->d_x8, 8, m->fe_W0t, m->fe_b0, Y, hid, dm, rows, 8, hid, 1));
    if (dp) BGNN_TRY(drop(Y, hid, dp->p_extractor, 1));
    BGNN_TRY(launch_gemm_f32(ctx, Y, hid, m->fe_W1t, m->fe_b1, X, hid, dm, rows, hid, hid, 0));
    float *dinv = asdX;
    if (d.gnn_type == BGNN_GNN_GCN) BGNN_TRY(launch_degree_inv_sqrt(ctx, g, dinv));
    const size_t nl = m->layers.size();
    // Eval mode on stencil graphs: a layer is ONE launch of the fused layer kernel in its plain-backbone mode -- aggregate ->
    // GEMM -> per-column post-op -- instead of reduce + GEMM launches with h round-tripping through HBM (GIN: its second Linear
    // stays a GEMM launch).  Anything the fused form does not cover falls through to the plain kernels below.
    const bool plain_fused = use_fused && g->kind == 0 && hid == 64;
    for (size_t l = 0; l < nl; ++l) {                 // invariant: X = h_l [rows][hid]
      const BgnnLayer &L = m->layers[l];
      const int relu = l + 1 < nl ? 1 : 0;
      if (plain_fused) {
        int rc = BGNN_ERR_UNSUPPORTED;
        if (d.gnn_type == BGNN_GNN_GCN) rc = launch_fused_plain_layer(ctx, g, 1, hid, X, dinv, L.Wfp, m->ones, L.scale, L.shift, relu, Y);
        else if (d.gnn_type == BGNN_GNN_SAGE) rc = launch_fused_plain_layer(ctx, g, 2, hid, X, nullptr, L.Wfp, m->ones, m->ones, L.b2, relu, Y);
        else rc = launch_fused_plain_layer(ctx, g, 3, hid, X, nullptr, L.Wfp, m->ones, m->ones, L.b1, 1, Y);
        if (rc == BGNN_OK) {
          if (d.gnn_type == BGNN_GNN_GIN) BGNN_TRY(launch_gemm_f32(ctx, Y, hid, L.Wt2, L.b2, X, hid, dm, rows, hid, hid, relu));
          else std::swap(X, Y);
          continue;
        }
        if (rc != BGNN_ERR_UNSUPPORTED) return rc;
      }
      if (d.gnn_type == BGNN_GNN_GCN) {               // lin, normalised aggregate, + bias, BatchNorm, ReLU
        BGNN_TRY(launch_gemm_f32(ctx, X, hid, L.Wt, nullptr, Y, hid, dm, rows, hid, hid, 0));
        if (tr) {
          BGNN_TRY(launch_neighbor_reduce(ctx, g, 1, Y, hid, dinv, m->ones, L.tr_bias, 0, X, hid, nullptr));
          BGNN_TRY(batch_norm(X, L, relu));
          if (dp && relu) BGNN_TRY(drop(X, hid, dp->p_features, 64 + (uint32_t)l));
        } else {
          BGNN_TRY(launch_neighbor_reduce(ctx, g, 1, Y, hid, dinv, L.scale, L.shift, relu, X, hid, nullptr));
        }
      } else if (d.gnn_type == BGNN_GNN_SAGE) {       // [mean_j x_j | x_i] @ [lin_l ; lin_r]^T (BatchNorm folded) + bias, ReLU
        BGNN_TRY(launch_neighbor_reduce(ctx, g, 2, X, hid, nullptr, nullptr, nullptr, 0, Y, 2 * hid, Y + hid));
        if (tr) {
          BGNN_TRY(launch_gemm_f32(ctx, Y, 2 * hid, L.tr_Wt, L.tr_bias, X, hid, dm, rows, 2 * hid, hid, 0));
          BGNN_TRY(batch_norm(X, L, relu));
          if (dp && relu) BGNN_TRY(drop(X, hid, dp->p_features, 64 + (uint32_t)l));
        } else {
          BGNN_TRY(launch_gemm_f32(ctx, Y, 2 * hid, L.Wt, L.b2, X, hid, dm, rows, 2 * hid, hid, relu));
        }
      } else {                                        // GIN: nn(sum_j x_j + x_i), nn = Linear ReLU Linear; BatchNorm; ReLU
        BGNN_TRY(launch_neighbor_reduce(ctx, g, 3, X, hid, nullptr, nullptr, nullptr, 0, Y, hid, nullptr));
        BGNN_TRY(launch_gemm_f32(ctx, Y, hid, L.Wt, L.b1, X, hid, dm, rows, hid, hid, 1));
        if (tr) {
          BGNN_TRY(launch_gemm_f32(ctx, X, hid, L.tr_Wt, L.tr_bias, Y, hid, dm, rows, hid, hid, 0));
          BGNN_TRY(batch_norm(Y, L, relu));
          if (dp && relu) BGNN_TRY(drop(Y, hid, dp->p_features, 64 + (uint32_t)l));
        } else {
          BGNN_TRY(launch_gemm_f32(ctx, X, hid, L.Wt2, L.b2, Y, hid, dm, rows, hid, hid, relu));
        }
        std::swap(X, Y);
      }
    }
    std::swap(X, Y);                                  // the tail below expects the backbone output in Y
  } else {
    const BgnnLayer &L0 = m->layers[0];
    if (ctx->opts.fold_extractor) {       // second extractor layer folded into lin_0 (see bgnn_model_create)
      const int sm = tr ? 0 : ctx->opts.matrix_path;   // training mode: exact float32 only (batch statistics amplify the split's error)
      const int smode = sm == 3 ? 3 : sm == 2 && m->l0f_Wsp16 ? 2 : sm ? 1 : 0;
      const float *wsplit = sm == 3 ? m->l0f_Wbf : sm == 2 && m->l0f_Wsp16 ? m->l0f_Wsp16 : sm ? m->l0f_Wsp : nullptr;
      // extractor layer 1 runs inside the lin_0 GEMM (same instructions, h1 never leaves the registers) wherever that GEMM takes
      // its W-resident form; below 32 768 rows (exact path) it keeps its own launch -- the results are bit-identical either way
      // (active extractor dropout sits between the two: the first layer then keeps its own launch)
      const bool front = hid == 64 && gemm_front_available(ctx, rows, L0.heads * hid, smode) && !(dp && dp->p_extractor > 0.0f);
      BGNN_REQUIRE(front || sm != 3, "matrix_path = bf16 needs fused_front = 1");
      // bf16 path, default shape: layer 0 aggregates the extractor's h1 and applies lin_0 afterwards, inside the fused launch
      // (gat_layer_bf16_2p_kernel, AF) -- no lin_0 product in HBM, no front GEMM
      if (sm == 3 && front && ctx->opts.bf16_layer0_af && ctx->opts.bf16_two_phase && m->l0af_W && nl_gat >= 2 && use_fused && !o->hidden) {
        const BgnnLayer &L1 = m->layers[1];
        const float *V3a = nullptr;
        if (g->kind == 0 && g->compact_edges && !g->edge_default) { const float *v3; BGNN_TRY(model_canonical_V(m, g, &v3)); V3a = v3; }
        if (L1.heads == 4 && L1.Wbf && g->kind == 0) {
          BGNN_TRY(launch_extractor_af(ctx, g->d_x8, m->fe_W0t, m->fe_b0, m->l0f_Wbf + (size_t)hid * L0.heads * hid / 2, Y, asdX, dm, rows, L0.heads));
          int rc = launch_fused_layer0_af(ctx, g, L0, L1, hid, V3a, Y, asdX, m->l0af_W, m->l0af_shift, X, asdY);
          if (rc == BGNN_OK) { std::swap(asdX, asdY); layer0_done = true; }
          else if (rc != BGNN_ERR_UNSUPPORTED) return rc;
        }
      }
      if (!layer0_done) {
      if (!front) BGNN_TRY(launch_gemm_f32(ctx, g->d_x8, 8, m->fe_W0t, m->fe_b0, Y, hid, dm, rows, 8, hid, 1));
      if (!front && dp) BGNN_TRY(drop(Y, hid, dp->p_extractor, 1));
      BGNN_TRY(launch_gemm_f32(ctx, front ? g->d_x8 : Y, front ? 8 : hid, m->l0f_Wt, m->l0f_b, X, L0.heads * hid, dm, rows, hid,
                               L0.heads * hid, 0, L0.att_src, L0.att_dst, asdX, L0.heads, hid, wsplit, smode,
                               front ? m->fe_W0t : nullptr, front ? m->fe_b0 : nullptr, front && smode == 0 ? m->l0f_Wpm : nullptr,
                               m->l0f_Wt_blk, smode == 2 ? m->l0f_Wsp16_inv : 1.0f));
      }
    } else {
      BGNN_REQUIRE(!bf16, "matrix_path = bf16 needs fold_extractor = 1");
      BGNN_TRY(launch_gemm_f32(ctx, g->d_x8, 8, m->fe_W0t, m->fe_b0, X, hid, dm, rows, 8, hid, 1));
      if (dp) BGNN_TRY(drop(X, hid, dp->p_extractor, 1));
      BGNN_TRY(launch_gemm_f32(ctx, X, hid, m->fe_W1t, m->fe_b1, Y, hid, dm, rows, hid, hid, 0));
      BGNN_TRY(launch_gemm_f32(ctx, Y, L0.d_in, L0.Wt, nullptr, X, L0.heads * hid, dm, rows, L0.d_in, L0.heads * hid, 0,
                               L0.att_src, L0.att_dst, asdX, L0.heads, hid, nullptr, 0, nullptr, nullptr, nullptr, L0.Wt_blk));
    }
  }
  // GNN backbone (gnn.py:173-188).  Invariant at the top of each iteration: X = lin_l(h_l), asdX = its dots.
  const size_t nl = gat ? m->layers.size() : 0;
  // a graph built with another edge feature list than the default: the fused kernels take the edge vectors over the canonical three
  const float *v3_all = nullptr;
  if (gat && use_fused && g->kind == 0 && g->compact_edges && !g->edge_default) BGNN_TRY(model_canonical_V(m, g, &v3_all));
  for (size_t l = layer0_done ? 1 : 0; l < nl; ++l) {
    const float *V3 = v3_all ? v3_all + l * (size_t)d.heads * 3 : nullptr;
    const BgnnLayer &Leval = m->layers[l];
    BgnnLayer Ltrain = Leval;                          // training mode: out = aggregate + bias, BatchNorm afterwards
    Ltrain.scale = m->ones; Ltrain.shift = Leval.tr_bias;
    const BgnnLayer &L = tr ? Ltrain : Leval;
    const int relu = L.concat ? 1 : 0;
    // GATConv(dropout = p) in training mode: the coefficients are thinned inside the plain aggregate kernel
    const bool att_drop = dp && dp->p_attention > 0.0f;
    const DropSpec att_spec = att_drop ? make_drop_spec(dp->p_attention, dp->seed, 16 + (uint32_t)l) : DropSpec{};
    if (l + 1 < nl) {
      const BgnnLayer &Ln = m->layers[l + 1];
      int rc = use_fused ? launch_fused_layer_next(ctx, g, L, Ln, hid, V3, X, asdX, Y, asdY) : BGNN_ERR_UNSUPPORTED;
      if (rc == BGNN_OK) { std::swap(X, Y); std::swap(asdX, asdY); continue; }
      if (rc != BGNN_ERR_UNSUPPORTED) return rc;
      BGNN_REQUIRE(!bf16, "matrix_path = bf16: no fused instance for layer %d of this model / graph", (int)l);
      rc = att_drop ? BGNN_ERR_UNSUPPORTED : launch_gat_aggregate_tiled(ctx, g, L, hid, d.edge_dim, X, asdX, Y, tr ? 0 : relu);
      if (rc == BGNN_ERR_UNSUPPORTED) rc = launch_gat_aggregate(ctx, g, L, hid, d.edge_dim, X, asdX, Y, tr ? 0 : relu, att_drop ? &att_spec : nullptr);
      BGNN_TRY(rc);
      if (tr) BGNN_TRY(batch_norm(Y, L, relu));
      if (dp && relu) BGNN_TRY(drop(Y, L.width, dp->p_features, 64 + (uint32_t)l));
      BGNN_TRY(launch_gemm_f32(ctx, Y, Ln.d_in, Ln.Wt, nullptr, X, Ln.heads * hid, dm, rows, Ln.d_in, Ln.heads * hid, 0,
                               Ln.att_src, Ln.att_dst, asdX, Ln.heads, hid, nullptr, 0, nullptr, nullptr, nullptr, Ln.Wt_blk));
    } else {
      int rc = use_fused ? launch_fused_layer_heads(ctx, g, m, L, hid, V3, X, asdX, thr_auto, thr_review,
                                                    grids ? grids->norm_floor : 0.01f, o, grids ? grids->cls : nullptr,
                                                    grids ? grids->conf : nullptr, grids ? grids->corr : nullptr)
                         : BGNN_ERR_UNSUPPORTED;
      if (rc == BGNN_OK) { if (grids) grids->done = true; return BGNN_OK; }
      if (rc != BGNN_ERR_UNSUPPORTED) return rc;
      BGNN_REQUIRE(!bf16, "matrix_path = bf16: no fused instance for the last layer of this model / graph");
      rc = att_drop ? BGNN_ERR_UNSUPPORTED : launch_gat_aggregate_tiled(ctx, g, L, hid, d.edge_dim, X, asdX, Y, tr ? 0 : relu);
      if (rc == BGNN_ERR_UNSUPPORTED) rc = launch_gat_aggregate(ctx, g, L, hid, d.edge_dim, X, asdX, Y, tr ? 0 : relu, att_drop ? &att_spec : nullptr);
      BGNN_TRY(rc);
      if (tr) BGNN_TRY(batch_norm(Y, L, relu));
      if (dp && relu) BGNN_TRY(drop(Y, L.width, dp->p_features, 64 + (uint32_t)l));     // (a single-layer backbone has no ReLU: never)
    }
  }
  if (o->hidden)                  // [N][logical hidden]: a padded model's pad columns (all zero) stay inside
    BGNN_TRY(launch_copy_cols(ctx, Y, hid, o->hidden, m->logical_hidden, m->logical_hidden, dm, rows));
  // heads (gnn.py:392-406)
  BGNN_TRY(launch_gemm_f32(ctx, Y, hid, m->hd_W0t, m->hd_b0, hidb, m->head_hidden_total, dm, rows, hid,
                           m->head_hidden_total, 1));
  if (dp && dp->p_heads > 0.0f)      // (the draw is indexed over the heads' own units; the table may carry pad columns up to a multiple of 32)
    BGNN_TRY(launch_dropout(ctx, hidb, head_count(&d) * (hid / 2), m->head_hidden_total, dm, rows, make_drop_spec(dp->p_heads, dp->seed, 2)));
  BGNN_TRY(launch_heads_final(ctx, m, hidb, m->head_hidden_total, dm, rows, thr_auto, thr_review, o));
  return BGNN_OK;
}

int bgnn_forward(bgnn_ctx *ctx, bgnn_model *m, bgnn_graph *g, float thr_auto, float thr_review, const bgnn_outputs *o) {
  BGNN_REQUIRE(ctx && m && g && o, "bgnn_forward: NULL argument");
  BGNN_REQUIRE(m->ctx == ctx && g->ctx == ctx, "bgnn_forward: model/graph belong to another context");
  BGNN_HIP_CHECK(hipSetDevice(ctx->device));
  return forward_impl(ctx, m, g, thr_auto, thr_review, o, nullptr);
}

// ---- sub-modules of the model on their own (parity tests against reference-generated fixtures) ----------------
static int row_count_on_device(bgnn_ctx *ctx, int64_t n, int64_t **d_n) {
  void *p;
  BGNN_TRY(ctx_workspace(ctx, 5, 64, &p));
  BGNN_TRY(ctx_upload(ctx, &n, sizeof(n), p));
  *d_n = (int64_t *)p;
  return BGNN_OK;
}

int bgnn_feature_extractor(bgnn_ctx *ctx, bgnn_model *m, const float *x, int64_t n_nodes, float *out) {
  BGNN_REQUIRE(ctx && m && out && (x || n_nodes == 0), "bgnn_feature_extractor: NULL argument");
  BGNN_REQUIRE(m->ctx == ctx, "bgnn_feature_extractor: model belongs to another context");
  BGNN_REQUIRE(n_nodes >= 0 && n_nodes < ((int64_t)1 << 30), "bgnn_feature_extractor: n_nodes=%lld out of range", (long long)n_nodes);
  if (n_nodes == 0) return BGNN_OK;
  BGNN_HIP_CHECK(hipSetDevice(ctx->device));
  const int hid = m->desc.hidden, in = m->desc.in_channels;
  void *px8, *ph;
  BGNN_TRY(ctx_workspace(ctx, 0, (size_t)n_nodes * 8 * sizeof(float), &px8));
  BGNN_TRY(ctx_workspace(ctx, 1, (size_t)n_nodes * hid * sizeof(float), &ph));
  int64_t *dn;
  BGNN_TRY(row_count_on_device(ctx, n_nodes, &dn));
  hipLaunchKernelGGL(pad_rows8_kernel, dim3((unsigned)((n_nodes * 8 + 255) / 256)), dim3(256), 0, ctx->stream, x, in,
                     (float *)px8, n_nodes);
  BGNN_HIP_CHECK(hipGetLastError());
  BGNN_TRY(launch_gemm_f32(ctx, (const float *)px8, 8, m->fe_W0t, m->fe_b0, (float *)ph, hid, dn, n_nodes, 8, hid, 1));
  if (!m->padded) return launch_gemm_f32(ctx, (const float *)ph, hid, m->fe_W1t, m->fe_b1, out, hid, dn, n_nodes, hid, hid, 0);
  void *po;                                               // padded width: out is [n][logical hidden]
  BGNN_TRY(ctx_workspace(ctx, 2, (size_t)n_nodes * hid * sizeof(float), &po));
  BGNN_TRY(launch_gemm_f32(ctx, (const float *)ph, hid, m->fe_W1t, m->fe_b1, (float *)po, hid, dn, n_nodes, hid, hid, 0));
  BGNN_TRY(launch_copy_cols(ctx, (const float *)po, hid, out, m->logical_hidden, m->logical_hidden, dn, n_nodes));
  return BGNN_OK;
}

int bgnn_heads(bgnn_ctx *ctx, bgnn_model *m, const float *hidden, int64_t n_nodes, float thr_auto, float thr_review,
               const bgnn_outputs *o) {
  BGNN_REQUIRE(ctx && m && o && (hidden || n_nodes == 0), "bgnn_heads: NULL argument");
  BGNN_REQUIRE(m->ctx == ctx, "bgnn_heads: model belongs to another context");
  BGNN_REQUIRE(n_nodes >= 0 && n_nodes < ((int64_t)1 << 30), "bgnn_heads: n_nodes=%lld out of range", (long long)n_nodes);
  BGNN_REQUIRE(!o->hidden, "bgnn_heads: `hidden` is this call's input");
  if (n_nodes == 0) return BGNN_OK;
  BGNN_HIP_CHECK(hipSetDevice(ctx->device));
  const int hid = m->desc.hidden;
  void *phid;
  BGNN_TRY(ctx_workspace(ctx, 3, (size_t)n_nodes * m->head_hidden_total * sizeof(float), &phid));
  int64_t *dn;
  BGNN_TRY(row_count_on_device(ctx, n_nodes, &dn));
  if (m->padded) {                                        // hidden is [n][logical hidden]: widen it with zero columns
    void *pw;
    BGNN_TRY(ctx_workspace(ctx, 1, (size_t)n_nodes * hid * sizeof(float), &pw));
    BGNN_TRY(launch_copy_cols(ctx, hidden, m->logical_hidden, (float *)pw, hid, m->logical_hidden, dn, n_nodes));
    hidden = (const float *)pw;
  }
  BGNN_TRY(launch_gemm_f32(ctx, hidden, hid, m->hd_W0t, m->hd_b0, (float *)phid, m->head_hidden_total, dn, n_nodes, hid,
                           m->head_hidden_total, 1));
  BGNN_TRY(launch_heads_final(ctx, m, (const float *)phid, m->head_hidden_total, dn, n_nodes, thr_auto, thr_review, o));
  return BGNN_OK;
}

int bgnn_forward_train(bgnn_ctx *ctx, bgnn_model *m, bgnn_graph *g, float *bn_batch_mean, float *bn_batch_var,
                       const bgnn_outputs *o) {
  return bgnn_forward_train_dropout(ctx, m, g, nullptr, bn_batch_mean, bn_batch_var, o);
}

int bgnn_forward_train_dropout(bgnn_ctx *ctx, bgnn_model *m, bgnn_graph *g, const bgnn_dropout *dropout, float *bn_batch_mean,
                               float *bn_batch_var, const bgnn_outputs *o) {
  BGNN_REQUIRE(ctx && m && g && o, "bgnn_forward_train: NULL argument");
  if (dropout) {
    const float ps[4] = {dropout->p_extractor, dropout->p_attention, dropout->p_features, dropout->p_heads};
    for (float p : ps) BGNN_REQUIRE(p >= 0.0f && p < 1.0f, "bgnn_forward_train_dropout: dropout probability %g outside [0, 1)", (double)p);
    if (ps[0] == 0.0f && ps[1] == 0.0f && ps[2] == 0.0f && ps[3] == 0.0f) dropout = nullptr;
  }
  BGNN_REQUIRE(m->ctx == ctx && g->ctx == ctx, "bgnn_forward_train: model/graph belong to another context");
  if (m->padded) {      // (batch statistics and dropout draws are laid out over the layer widths the caller sees)
    set_error("bgnn_forward_train: hidden_channels=%d / heads=%d run zero-padded to %d / %d; the training-mode forward exists for "
              "hidden 32 / 64 / 128 and power-of-two head counts only", m->logical_hidden, m->logical_heads, m->desc.hidden, m->desc.heads);
    return BGNN_ERR_UNSUPPORTED;
  }
  BGNN_REQUIRE(!o->action && !o->needs_review && !o->auto_correct, "bgnn_forward_train: the deployment flags belong to predict()");
  BGNN_HIP_CHECK(hipSetDevice(ctx->device));
  int64_t c[4];
  BGNN_HIP_CHECK(hipMemcpyAsync(c, g->d_counts, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
  BGNN_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  // torch.nn.functional.batch_norm in training mode refuses a single row the same way
  BGNN_REQUIRE(c[0] != 1, "Expected more than 1 value per channel when training, got input size [1, %d]", m->layers[0].width);
  TrainOut tr{bn_batch_mean, bn_batch_var, dropout};
  return forward_impl(ctx, m, g, 0.85f, 0.6f, o, nullptr, &tr);
}

int bgnn_infer_tiles(bgnn_ctx *ctx, bgnn_model *m, const bgnn_tiles *tiles, const bgnn_graph_opts *opts, float thr_auto,
                     float thr_review, float norm_floor, float *classification, float *confidence, float *correction,
                     int64_t *n_nodes_out) {
  BGNN_REQUIRE(ctx && m && tiles && opts, "bgnn_infer_tiles: NULL argument");
  bgnn_graph *g = nullptr;
  float *const grids[3] = {classification, confidence, correction};
  // (the compaction scan writes the node count there itself; a ragged batch's canvas fill zero-fills the result grids on its way)
  BGNN_TRY(graph_build_impl(ctx, tiles, opts, &g, n_nodes_out, grids));
  const int64_t rows = g->row_capacity;
  GridOut go;
  go.cls = classification; go.conf = confidence; go.corr = correction; go.norm_floor = norm_floor;
  // first try the fully fused tail (no per-node outputs at all); otherwise per-node outputs + K6
  void *p;
  int rc = ctx_workspace(ctx, 4, (size_t)rows * (sizeof(int64_t) + 2 * sizeof(float)), &p);
  if (rc == BGNN_OK) {
    bgnn_outputs none{};
    bgnn_outputs o{};
    o.predicted_class = (int64_t *)p;
    o.confidence = (float *)(o.predicted_class + rows);
    o.correction = m->desc.predict_correction ? o.confidence + rows : nullptr;
    const bool try_fused = fused_heads_available(ctx, g, m);
    if (try_fused && g->d_atlas && !g->grids_cleared) {   // the canvas walk writes valid cells only: clear the grids (fill 0.0) first
      const size_t nb = (size_t)g->total_cells * sizeof(float);
      if (classification && confidence == classification + g->total_cells && correction == confidence + g->total_cells) {
        BGNN_HIP_CHECK(hipMemsetAsync(classification, 0, 3 * nb, ctx->stream));      // one [3, cells] block: one fill
      } else {
        if (classification) BGNN_HIP_CHECK(hipMemsetAsync(classification, 0, nb, ctx->stream));
        if (confidence) BGNN_HIP_CHECK(hipMemsetAsync(confidence, 0, nb, ctx->stream));
        if (correction) BGNN_HIP_CHECK(hipMemsetAsync(correction, 0, nb, ctx->stream));
      }
    }
    rc = forward_impl(ctx, m, g, thr_auto, thr_review, try_fused ? &none : &o, &go);
    if (rc == BGNN_OK && !go.done) {
      if (try_fused) rc = forward_impl(ctx, m, g, thr_auto, thr_review, &o, nullptr);   // (not reached in practice)
      if (rc == BGNN_OK)
        rc = launch_results_to_grids(g, o.predicted_class, o.confidence, o.correction, norm_floor, classification,
                                     confidence, correction);
    }
  }
  graph_free(g);   // buffers return to the pool; stream order keeps them valid for the work already queued
  return rc;
}

}  // extern "C"
