// VR BAG refinement records on the device: the two steps either side of bgnn_infer_tiles when a
// whole `varres_refinements` array is resident in HBM.
//
// The array is a run of {depth, depth_uncrt} float32 records, every refinement grid's cells row-major and the
// grids one after another in `varres_metadata.index` order (reference data/vr_bag.py:262-276) -- which IS the
// concatenated tile layout bgnn_tiles wants, so no host packing is needed:
//   bgnn_vr_unpack : records -> depth / uncertainty planes, valid mask (RefinementGrid.valid_mask, vr_bag.py:88-92),
//                    per-grid valid counts and the min_valid_ratio filter of iterate_refinements (:293-295)
//   bgnn_vr_apply  : the write-back arithmetic of scripts/inference_native.py:480-503 (apply_results) on the records,
//                    in place, with the counters its log prints
// HBM-bound byte work: 8 B records in, 9 B out (unpack); 25 B in, 8 B out (apply).  -ffp-contract=off: the
// uncertainty scale is rounded to float32 before the product, as numpy does.
#include "bgnn_internal.h"

namespace bgnn {

struct VrUnpackArgs {
  const float2 *rec;
  int64_t n;
  float nodata;
  float *depth, *unc;   // unc may be NULL
  uint8_t *mask;
};

__global__ __launch_bounds__(256) void vr_unpack_kernel(VrUnpackArgs a) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
    const float2 r = a.rec[i];
    a.depth[i] = r.x;
    if (a.unc) a.unc[i] = r.y;
    a.mask[i] = (r.x != a.nodata) && (fabsf(r.x) <= 3.402823466e+38f);   // != nodata & isfinite (NaN fails <=)
  }
}

struct VrFilterArgs {
  int32_t n_grids;
  const int64_t *cell_off;   // [n_grids + 1]
  double min_valid_ratio;
  uint8_t *mask;
  int64_t *valid_count;      // [n_grids]
  uint8_t *keep;             // [n_grids]
};

// one wave per refinement grid: count its valid cells; a grid below the ratio is dropped from the batch by
// clearing its mask (an all-invalid grid yields all-zero results, what add_to_batch returns for it, :262-264)
__global__ __launch_bounds__(256) void vr_filter_kernel(VrFilterArgs a) {
  const int lane = threadIdx.x & 63;
  const int g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (g >= a.n_grids) return;
  const int64_t lo = a.cell_off[g], hi = a.cell_off[g + 1];
  int cnt = 0;
  for (int64_t i = lo + lane; i < hi; i += 64) cnt += a.mask[i];
  for (int o = 32; o; o >>= 1) cnt += __shfl_xor(cnt, o);
  // Python: grid.num_valid / grid.depth.size >= min_valid_ratio, both sides float64
  const bool keep = hi > lo && ((double)cnt / (double)(hi - lo)) >= a.min_valid_ratio;
  if (!keep)
    for (int64_t i = lo + lane; i < hi; i += 64) a.mask[i] = 0;
  if (lane == 0) { a.valid_count[g] = cnt; a.keep[g] = keep; }
}

struct VrApplyArgs {
  float2 *rec;
  int64_t n;
  const uint8_t *mask;
  const float *cls, *conf, *corr;
  float thr;
  unsigned long long *counts;   // [0] noise & valid, [1] corrected, [2] depth actually changed
  double *conf_sum;             // sum of confidence over valid cells
};

__global__ __launch_bounds__(256) void vr_apply_kernel(VrApplyArgs a) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned n_noise = 0, n_corr = 0, n_chg = 0;
  double csum = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
    const bool valid = a.mask[i] != 0;
    const float conf = a.conf[i];
    const bool noise = valid && a.cls[i] == 2.0f;            // CLASS_NOISE
    if (valid) csum += (double)conf;
    n_noise += noise;
    if (noise && conf >= a.thr) {                            // '>=' here (inference_native.py:489), '>' in the tiled pipeline
      float2 r = a.rec[i];
      const float d = r.x - a.corr[i];
      const float scale = 2.0f - conf;
      n_chg += d != r.x;
      r.x = d;
      r.y = r.y * scale;
      a.rec[i] = r;
      ++n_corr;
    }
  }
  // wave, then block, then one atomic per block and counter
  for (int o = 32; o; o >>= 1) {
    n_noise += __shfl_xor(n_noise, o); n_corr += __shfl_xor(n_corr, o); n_chg += __shfl_xor(n_chg, o);
    csum += __shfl_xor(csum, o);
  }
  __shared__ unsigned s_cnt[4][3];
  __shared__ double s_sum[4];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_cnt[w][0] = n_noise; s_cnt[w][1] = n_corr; s_cnt[w][2] = n_chg; s_sum[w] = csum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long c0 = 0, c1 = 0, c2 = 0;
    double s = 0.0;
    for (int k = 0; k < 4; ++k) { c0 += s_cnt[k][0]; c1 += s_cnt[k][1]; c2 += s_cnt[k][2]; s += s_sum[k]; }
    if (c0) atomicAdd(&a.counts[0], c0);
    if (c1) atomicAdd(&a.counts[1], c1);
    if (c2) atomicAdd(&a.counts[2], c2);
    if (s != 0.0) atomicAdd(a.conf_sum, s);
  }
}

static int grid_for(int64_t n) {
  const int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace bgnn

using namespace bgnn;

extern "C" int bgnn_vr_unpack(bgnn_ctx *ctx, const float *records, int64_t n_cells, float nodata, int32_t n_grids,
                              const int64_t *cell_offsets, double min_valid_ratio, float *depth, float *uncertainty,
                              uint8_t *mask, int64_t *valid_count, uint8_t *keep) {
  BGNN_REQUIRE(ctx && records && depth && mask, "bgnn_vr_unpack: NULL argument");
  BGNN_REQUIRE(n_cells >= 0 && n_grids >= 0, "bgnn_vr_unpack: bad sizes");
  BGNN_REQUIRE(n_grids == 0 || (cell_offsets && valid_count && keep), "bgnn_vr_unpack: grid table without outputs");
  BGNN_HIP_CHECK(hipSetDevice(ctx->device));
  ProfScope ps(ctx, BGNN_K_SCATTER);
  if (n_cells > 0) {
    VrUnpackArgs a{reinterpret_cast<const float2 *>(records), n_cells, nodata, depth, uncertainty, mask};
    hipLaunchKernelGGL(vr_unpack_kernel, dim3(grid_for(n_cells)), dim3(256), 0, ctx->stream, a);
  }
  if (n_grids > 0) {
    VrFilterArgs f{n_grids, cell_offsets, min_valid_ratio, mask, valid_count, keep};
    hipLaunchKernelGGL(vr_filter_kernel, dim3((n_grids + 3) / 4), dim3(256), 0, ctx->stream, f);
  }
  BGNN_HIP_CHECK(hipGetLastError());
  return BGNN_OK;
}

extern "C" int bgnn_vr_apply(bgnn_ctx *ctx, float *records, int64_t n_cells, const uint8_t *mask,
                             const float *classification, const float *confidence, const float *correction,
                             float auto_correct_threshold, uint64_t *counts, double *confidence_sum) {
  BGNN_REQUIRE(ctx && records && mask && classification && confidence && correction && counts && confidence_sum,
               "bgnn_vr_apply: NULL argument");
  BGNN_REQUIRE(n_cells >= 0, "bgnn_vr_apply: bad size");
  BGNN_HIP_CHECK(hipSetDevice(ctx->device));
  if (n_cells == 0) return BGNN_OK;
  ProfScope ps(ctx, BGNN_K_SCATTER);
  VrApplyArgs a{reinterpret_cast<float2 *>(records), n_cells, mask, classification, confidence, correction,
                auto_correct_threshold, reinterpret_cast<unsigned long long *>(counts), confidence_sum};
  hipLaunchKernelGGL(vr_apply_kernel, dim3(grid_for(n_cells)), dim3(256), 0, ctx->stream, a);
  BGNN_HIP_CHECK(hipGetLastError());
  return BGNN_OK;
}
