// Survey <-> tiles on the device: stitching of overlapping tile results into survey grids, and
// (bottom of the file) cutting tiles out of a survey resident in HBM.
//
// Restates TileMerger.add_tile / finalize (reference data/tiling.py:384-454), TileManager.merge_tile /
// finalize_output (:218-294), the unprocessed-cell preservation of BathymetricPipeline.process
// (models/pipeline.py:196-207) and _apply_corrections (:316-349) as ONE gather kernel: each survey cell
// walks the (<= 3 x 3) tiles that cover it in ascending spec order -- the order the reference's serial
// loop adds them in -- so the float32 blend sums and the strict '>' confidence arbitration come out
// identical to the host merge.  Compiled with -ffp-contract=off (product rounded, then added, as numpy).
#include "bgnn_internal.h"

namespace bgnn {

struct StitchArgs {
  int H, W;
  int ntr, ntc;                 // tile rows / cols
  const int32_t *row_start;     // [ntr]
  const int32_t *row_end;       // [ntr]
  const int32_t *col_start;     // [ntc]
  const int32_t *col_end;       // [ntc]
  const float *roww;            // [ntr][tile]  1-D blend weights of each tile row's extent
  const float *colw;            // [ntc][tile]
  int tile;                     // pitch of the two weight tables
  const int64_t *tile_off;      // [ntr*ntc] offset of the tile's cells in the result arrays; < 0: tile skipped
  const float *cls, *conf, *corr;   // concatenated per-tile grids (row-major [h][w] each)
  const float *depth;           // [H][W] survey depth
  const uint8_t *valid;         // [H][W] survey valid mask
  float auto_thr;
  float *o_cls, *o_conf, *o_corr, *o_clean;   // [H][W]
};

__global__ __launch_bounds__(256) void stitch_kernel(StitchArgs a) {
  const int nbx = (a.W + 255) / 256;                     // linear grid: surveys taller than 65 535 rows fit too
  const int r = blockIdx.x / nbx;
  const int c = (blockIdx.x - r * nbx) * blockDim.x + threadIdx.x;
  if (c >= a.W) return;
  const float NANF = __builtin_nanf("");
  float acc_conf = NANF, w_conf = 0.0f, acc_corr = NANF, w_corr = 0.0f;
  float lab = NANF, tracked = -1.0f;
  // candidate tile rows / cols: starts and ends are ascending, so bisect to the first extent that ends past the
  // cell and walk the few that contain it, in ascending order
  int tr0 = 0, tc0 = 0;
  for (int lo = 0, hi = a.ntr; lo < hi;) { const int m = (lo + hi) >> 1; if (a.row_end[m] > r) hi = m; else lo = m + 1; tr0 = lo; }
  for (int lo = 0, hi = a.ntc; lo < hi;) { const int m = (lo + hi) >> 1; if (a.col_end[m] > c) hi = m; else lo = m + 1; tc0 = lo; }
  for (int tr = tr0; tr < a.ntr; ++tr) {
    const int rs = a.row_start[tr];
    if (rs > r) break;
    if (r >= a.row_end[tr]) continue;
    for (int tc = tc0; tc < a.ntc; ++tc) {
      const int cs = a.col_start[tc];
      if (cs > c) break;
      if (c >= a.col_end[tc]) continue;
      const int64_t off = a.tile_off[tr * a.ntc + tc];
      if (off < 0) continue;                                    // tile skipped (min_valid_ratio)
      const int tw = a.col_end[tc] - cs;
      const int64_t i = off + (int64_t)(r - rs) * tw + (c - cs);
      const float w = a.roww[tr * a.tile + (r - rs)] * a.colw[tc * a.tile + (c - cs)];   // np.outer, float32
      const float vconf = a.conf[i], vcorr = a.corr[i], vcls = a.cls[i];
      // continuous channels: weighted accumulation (tiling.py:242-258)
      if (vconf == vconf && fabsf(vconf) != __builtin_inff()) {
        if (acc_conf != acc_conf) acc_conf = 0.0f;
        w_conf = w_conf + w;
        acc_conf = acc_conf + vconf * w;
      }
      if (vcorr == vcorr && fabsf(vcorr) != __builtin_inff()) {
        if (acc_corr != acc_corr) acc_corr = 0.0f;
        w_corr = w_corr + w;
        acc_corr = acc_corr + vcorr * w;
      }
      // discrete channel: the tile with the higher confidence wins; first writer always (tiling.py:408-420)
      if (vcls == vcls && fabsf(vcls) != __builtin_inff()) {
        if (vconf > tracked || lab != lab) { lab = vcls; tracked = vconf; }
      }
    }
  }
  if (w_conf > 0.0f) acc_conf = acc_conf / w_conf;              // finalize_output (tiling.py:289-292)
  if (w_corr > 0.0f) acc_corr = acc_corr / w_corr;
  const int64_t o = (int64_t)r * a.W + c;
  const bool valid = a.valid[o] != 0;
  if (valid && lab != lab) { lab = 0.0f; acc_conf = 0.0f; acc_corr = 0.0f; }   // pipeline.py:198-207
  const float d = a.depth[o];
  float clean = d;
  if (lab == 2.0f && acc_conf > a.auto_thr && valid) clean = d - acc_corr;      // pipeline.py:337-342
  a.o_cls[o] = lab; a.o_conf[o] = acc_conf; a.o_corr[o] = acc_corr; a.o_clean[o] = clean;
}

// ---- the other direction: cutting tiles out of a survey that is resident in HBM ---------------------------------
// TileManager.extract_tile / iterate_tiles (reference data/tiling.py:136-216) for a batch of equally sized tiles:
// windows of the survey's depth / valid mask / uncertainty are copied into the concatenated tile layout
// bgnn_infer_tiles consumes.  5 (9) B read + written per cell: HBM-trivial next to the forward.
struct CutArgs {
  int W;                        // survey row pitch (cells)
  int th, tw;
  const int32_t *origin;        // [n_tiles][2] (row_start, col_start)
  const float *depth; const uint8_t *valid; const float *unc;
  float *o_depth; uint8_t *o_valid; float *o_unc;
};

__global__ __launch_bounds__(256) void cut_tiles_kernel(CutArgs a) {
  const int t = blockIdx.z, r = blockIdx.y;
  const int r0 = a.origin[2 * t], c0 = a.origin[2 * t + 1];
  const int64_t src = (int64_t)(r0 + r) * a.W + c0;
  const int64_t dst = ((int64_t)t * a.th + r) * a.tw;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < a.tw; c += gridDim.x * blockDim.x) {
    a.o_depth[dst + c] = a.depth[src + c];
    a.o_valid[dst + c] = a.valid[src + c];
    if (a.unc) a.o_unc[dst + c] = a.unc[src + c];
  }
}

struct CountArgs {
  int W, th, tw;
  const int32_t *origin;
  const uint8_t *valid;
  int64_t *count;               // [n_tiles]
};

// one workgroup per tile: number of valid cells in its window (iterate_tiles' valid_ratio test, tiling.py:203-209)
__global__ __launch_bounds__(256) void tile_valid_count_kernel(CountArgs a) {
  const int t = blockIdx.x;
  const int r0 = a.origin[2 * t], c0 = a.origin[2 * t + 1];
  unsigned cnt = 0;
  for (int i = threadIdx.x; i < a.th * a.tw; i += 256) {
    const int r = i / a.tw, c = i - r * a.tw;
    cnt += a.valid[(int64_t)(r0 + r) * a.W + c0 + c] != 0;
  }
  for (int o = 32; o; o >>= 1) cnt += __shfl_xor(cnt, o);
  __shared__ unsigned s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) a.count[t] = (int64_t)s[0] + s[1] + s[2] + s[3];
}

}  // namespace bgnn

using namespace bgnn;

extern "C" int bgnn_cut_tiles(bgnn_ctx *ctx, int32_t height, int32_t width, const float *depth, const uint8_t *valid_mask,
                              const float *uncertainty, int32_t n_tiles, const int32_t *origins, int32_t tile_h,
                              int32_t tile_w, float *out_depth, uint8_t *out_mask, float *out_uncertainty) {
  BGNN_REQUIRE(ctx && depth && valid_mask && origins && out_depth && out_mask, "bgnn_cut_tiles: NULL argument");
  BGNN_REQUIRE((uncertainty == nullptr) == (out_uncertainty == nullptr), "bgnn_cut_tiles: uncertainty in and out must both be given or both be NULL");
  BGNN_REQUIRE(height > 0 && width > 0 && tile_h > 0 && tile_w > 0 && tile_h <= height && tile_w <= width && n_tiles >= 0 &&
                   tile_h <= 65535 && n_tiles <= 65535, "bgnn_cut_tiles: bad sizes");
  BGNN_HIP_CHECK(hipSetDevice(ctx->device));
  if (n_tiles == 0) return BGNN_OK;
  ProfScope ps(ctx, BGNN_K_SCATTER);
  CutArgs a{width, tile_h, tile_w, origins, depth, valid_mask, uncertainty, out_depth, out_mask, out_uncertainty};
  hipLaunchKernelGGL(cut_tiles_kernel, dim3((tile_w + 255) / 256 > 4 ? 4 : (tile_w + 255) / 256, tile_h, n_tiles), dim3(256), 0,
                     ctx->stream, a);
  BGNN_HIP_CHECK(hipGetLastError());
  return BGNN_OK;
}

extern "C" int bgnn_tile_valid_counts(bgnn_ctx *ctx, int32_t height, int32_t width, const uint8_t *valid_mask, int32_t n_tiles,
                                      const int32_t *origins, int32_t tile_h, int32_t tile_w, int64_t *counts) {
  BGNN_REQUIRE(ctx && valid_mask && origins && counts, "bgnn_tile_valid_counts: NULL argument");
  BGNN_REQUIRE(height > 0 && width > 0 && tile_h > 0 && tile_w > 0 && tile_h <= height && tile_w <= width && n_tiles >= 0,
               "bgnn_tile_valid_counts: bad sizes");
  BGNN_HIP_CHECK(hipSetDevice(ctx->device));
  if (n_tiles == 0) return BGNN_OK;
  ProfScope ps(ctx, BGNN_K_SCATTER);
  CountArgs a{width, tile_h, tile_w, origins, valid_mask, counts};
  hipLaunchKernelGGL(tile_valid_count_kernel, dim3(n_tiles), dim3(256), 0, ctx->stream, a);
  BGNN_HIP_CHECK(hipGetLastError());
  return BGNN_OK;
}

extern "C" int bgnn_stitch_tiles(bgnn_ctx *ctx, int32_t height, int32_t width, int32_t n_tile_rows, int32_t n_tile_cols,
                                 const int32_t *row_start, const int32_t *row_end, const int32_t *col_start,
                                 const int32_t *col_end, const float *row_weights, const float *col_weights,
                                 int32_t weight_pitch, const int64_t *tile_offsets, const float *classification,
                                 const float *confidence, const float *correction, const float *depth,
                                 const uint8_t *valid_mask, float auto_correct_threshold, float *out_classification,
                                 float *out_confidence, float *out_correction, float *out_cleaned_depth) {
  BGNN_REQUIRE(ctx && row_start && row_end && col_start && col_end && row_weights && col_weights && tile_offsets &&
                   classification && confidence && correction && depth && valid_mask && out_classification &&
                   out_confidence && out_correction && out_cleaned_depth,
               "bgnn_stitch_tiles: NULL argument");
  BGNN_REQUIRE(height > 0 && width > 0 && n_tile_rows > 0 && n_tile_cols > 0 && weight_pitch > 0, "bgnn_stitch_tiles: bad sizes");
  BGNN_HIP_CHECK(hipSetDevice(ctx->device));
  StitchArgs a{height, width, n_tile_rows, n_tile_cols, row_start, row_end, col_start, col_end, row_weights, col_weights,
               weight_pitch, tile_offsets, classification, confidence, correction, depth, valid_mask,
               auto_correct_threshold, out_classification, out_confidence, out_correction, out_cleaned_depth};
  ProfScope ps(ctx, BGNN_K_SCATTER);
  const int64_t n_blocks = (int64_t)((width + 255) / 256) * height;
  BGNN_REQUIRE(n_blocks < ((int64_t)1 << 31), "bgnn_stitch_tiles: survey too large for one launch (%lld workgroups)", (long long)n_blocks);
  hipLaunchKernelGGL(stitch_kernel, dim3((unsigned)n_blocks), dim3(256), 0, ctx->stream, a);
  BGNN_HIP_CHECK(hipGetLastError());
  return BGNN_OK;
}
