// Internal structures shared by the translation units of libbgnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <cstdlib>
#include <vector>
#include <map>
#include <set>
#include <atomic>

#include "../../include/bgnn.h"

namespace bgnn {

void set_error(const char *fmt, ...);

#define BGNN_HIP_CHECK(expr)                                                          \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      bgnn::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                      __LINE__);                                                      \
      return BGNN_ERR_HIP;                                                            \
    }                                                                                 \
  } while (0)

#define BGNN_REQUIRE(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      bgnn::set_error(__VA_ARGS__);    \
      return BGNN_ERR_INVALID;         \
    }                                  \
  } while (0)

#define BGNN_TRY(expr)          \
  do {                          \
    int _r = (expr);            \
    if (_r != BGNN_OK) return _r; \
  } while (0)

// ---- device memory pool: size-keyed free lists, so steady-state batches never hipMalloc --
struct DevPool {
  std::multimap<size_t, void *> free_blocks;
  std::map<void *, size_t> live;
  size_t total_bytes = 0;
  int alloc(size_t bytes, void **out);
  void release(void *p);
  void trim();
};

struct ProfRecord {
  int kernel;
  hipEvent_t start, stop;
};

}  // namespace bgnn

// Per-tile metadata (device + host copy)
struct BgnnTileMeta {
  int32_t h, w;
  int32_t cell_off;   // first cell of the tile in the concatenated cell space
  int32_t pad;
  double rx, ry;      // resolution (x, y)
};

// Work item of the tile-structured kernels: a band of rows of one tile
struct BgnnWorkItem {
  int32_t tile, r0, nr, pad;
};

// Run-time switches of a context.  Defaults come from the environment ONCE, at bgnn_ctx_create (BGNN_SPLIT_F16 /
// BGNN_SPLIT_BF16, BGNN_NO_FUSED, BGNN_NO_FOLD, ...); afterwards only bgnn_ctx_set_option changes them -- no getenv on
// the launch path.
struct BgnnOpts {
  int matrix_path = 0;       // 0 exact f32, 1 bf16x3, 2 fp16x3 (opt-in operand-split matrix paths), 3 bf16 activation storage + bf16 MFMA
  int fused = 1;             // 0: K3 / K4 / K5 / K6 as separate kernels
  int fold_extractor = 1;    // 0: run the extractor's second Linear and lin of layer 0 unfolded
  int fused_persistent = 0;  // 1 (opt-in experiment): big uniform batches run the 256 -> 256 exact-f32 fused layer in its persistent
                             // one-workgroup-per-CU form (bit-identical; measured 11.4 ms per launch against 10.15: DESIGN.md)
  int bf16_two_phase = 1;    // matrix_path = bf16: the 256 -> 256 fused layer in its two-phase form (aggregate all slabs to bf16 registers, then
                             // the GEMM in four column passes: three workgroups per CU; bit-identical to the one-phase instance, 0 selects that;
                             // 2: the 256 -> 64 instance in the same form too -- experiment, neutral)
  int bf16_layer0_af = 1;    // matrix_path = bf16, default model shape: layer 0 aggregates the extractor's 64-channel h1 and applies the folded lin_0
                             // weight per head afterwards, inside the fused launch (no front GEMM, no 512-byte lin_0 rows in HBM); 0: front GEMM + the
                             // ordinary two-phase launch.  Same mathematics, another rounding sequence (not bit-identical to 0)
  int stats_narrow = -1;     // box statistics with 16 instead of 64 running sums per workgroup (four times the workgroups, a quarter of the
                             // work between two barriers): -1 picks it when the wide launch would leave most CUs idle; bit-identical
  int fused_front = 1;       // 1: extractor layer 1 runs inside the lin_0 GEMM where that GEMM's W-resident form is used (0: own launch)
  int features_tiled = 1;    // 1: LDS-tiled feature kernel with mirrored-edge slope reuse (K = 8 / 16); 0: thread-per-cell form (bit-identical)
  int ragged_atlas = 1;      // ragged batches: fused layers walk a shelf-packed canvas of the grids (0: per-grid 8x16 blocks)
  int fused_lds_pad_kb = 0;    // experiment: pad the fused kernel's LDS request (occupancy)
  int diag_mask = 0;         // BGNN_DIAG builds only: phase ablation bits of the fused kernel
  int diag_stamps = 0;       // BGNN_DIAG builds only: per-phase s_memtime sums
  int gemm_waves = 8, gemm_diag = 0, gemm_no_wres = 0;
  int gemm_pair_major = 1;   // exact-f32 lin_0 GEMM with the extractor in front: tile-pair-major MFMA order, the epilogue of pair p (bias, attention
                             // dots, transposed row stores) issued between the MFMAs of pair p + 1 (0: tile-major, epilogue after; bit-identical)
};

// Diagnostics (phase ablations, cycle stamps) are compiled in only with -DBGNN_DIAG=1 (python __graft_entry__.py --diag
// builds libbgnn_hip_diag.so); the production kernels carry none of it.
#ifndef BGNN_DIAG
#define BGNN_DIAG 0
#endif

struct bgnn_ctx {
  int device = 0;
  BgnnOpts opts;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  bgnn::DevPool pool;
  // profiling
  uint32_t prof_mask = 0;
  std::vector<bgnn::ProfRecord> prof_records;
  std::vector<hipEvent_t> event_pool;
  // forward workspace (grow-only)
  void *ws[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t ws_bytes[6] = {0, 0, 0, 0, 0, 0};
  // pinned staging for host->device tables: a table is copied into a pinned buffer at call time and DMA'd from there,
  // so the caller's (often function-local, pageable) source may go away while the copy is still queued behind
  // other stream work; a buffer is reused once its event has completed
  struct Staging { void *p; size_t cap; hipEvent_t ev; bool in_flight; };
  std::vector<Staging> staging;
  // tile / work-item tables of uniform batches, kept on the device across calls: a pipeline cuts thousands of batches of one
  // shape, and a single-tile call should not pay two host->device copies for 4 KiB of tables
  // Entries are reference-counted by the graphs that point into them (bgnn_graph::table_cache_id): an entry is evicted only
  // when no live graph uses it; a graph that finds the cache full of pinned entries gets private pool tables instead.
  struct TableCache { int32_t n_tiles, h, w, item_cells; double rx, ry; BgnnTileMeta *d_tiles; BgnnWorkItem *d_items; int32_t n_items; uint64_t stamp; uint64_t id; int32_t refs; };
  std::vector<TableCache> table_cache;
  uint64_t table_stamp = 0, table_next_id = 1;
  // graphs built on this context and not yet destroyed: bgnn_ctx_destroy releases them (their handles die with the context)
  std::set<struct bgnn_graph *> live_graphs;
  int num_cus = 256;
  float *zero_page = nullptr;   // 16 KiB: [0,4K) zeros, [4K,4K+128) diagnostic counters, [8K,16K) dump rows
  unsigned long long *stamps = nullptr;   // 16 diagnostic counters (inside the zero page allocation)
};

struct BgnnLayer {
  int d_in, heads, width;   // width = heads*hidden (concat) or hidden (last)
  int concat;
  float *Wt;        // [d_in][heads*hidden]   (lin.weight transposed)
  float *Wt_blk = nullptr;   // layers wider than 256 columns: Wt as [columns / 256][d_in][256] (the GEMM runs one launch per 256 columns)
  float *att_src;   // [heads*hidden]
  float *att_dst;   // [heads*hidden]
  float *V;         // [heads][edge_dim]   folded lin_edge . att_edge
  float *scale;     // [width]  BN weight / sqrt(var + eps)
  float *shift;     // [width]  (conv bias - mean) * scale + BN bias
  float *Wsp;       // Wt as a bf16 hi / lo split image for the bf16x3 matrix path (same byte geometry as Wt; see pack_split)
  float *Wsp16;     // the same with float16 parts (fp16x3), of W * 2^S ...
  float Wsp16_inv = 1.0f;   // ... 2^-S: the kernels' accumulators are multiplied by it (bgnn_api.hip pack_split)
  float *Wfp;       // Wt with the columns of every row permuted for the fused exact-f32 kernel (gat_layer_fused.hip: WTileGroup)
  float *Wbf;       // Wt as a bf16 (hi only) image for the bf16 storage path: [D/16][NC/32][1 KiB] in MFMA A-fragment lane order
  // non-attention backbones (desc.gnn_type != BGNN_GNN_GAT): Wt = GCN lin^T [hid][hid] | SAGE [lin_l^T ; lin_r^T] [2 hid][hid]
  // with BatchNorm folded in | GIN nn.0^T [hid][hid]; then
  float *b1;        // GIN nn.0 bias [hid]
  float *Wt2;       // GIN nn.2^T [hid][hid] with BatchNorm folded in
  float *b2;        // SAGE / GIN: bias with BatchNorm folded in [hid]
  // unfolded pieces for the training-mode forward (BatchNorm statistics taken from the batch, bn_train.hip)
  float *tr_bias;   // the convolution's own bias [width] (GAT bias, GCN bias, SAGE lin_l bias, GIN nn.2 bias)
  float *bn_w, *bn_b;   // BatchNorm weight / bias [width]
  float *tr_Wt;     // SAGE [lin_l^T ; lin_r^T], GIN nn.2^T without the BatchNorm fold (else nullptr)
};

struct bgnn_model {
  bgnn_ctx *ctx;
  bgnn_model_desc desc;       // the shape the KERNELS run: hidden / heads zero-padded to a supported width (bgnn_model_create)
  int logical_hidden = 0, logical_heads = 0;   // the caller's shape: widths of `hidden` in / out, bgnn_model_weight_count
  bool padded = false;
  float *blob = nullptr;      // one device allocation holding everything below
  size_t blob_floats = 0;
  float *fe_W0t, *fe_b0, *fe_W1t, *fe_b1;     // [in8][hid], [hid], [hid][hid], [hid]
  float *l0f_Wt, *l0f_b;      // [hid][HC0], [HC0]: second extractor layer folded into lin of layer 0 (no activation between)
  float *l0f_Wsp = nullptr, *l0f_Wsp16 = nullptr;   // l0f_Wt as bf16 / float16 hi / lo split images
  float *l0f_Wbf = nullptr, *hd_W0bf = nullptr;     // l0f_Wt / hd_W0t as bf16 (hi only) images
  float *l0f_Wpm = nullptr;                         // l0f_Wt with the columns of tile pairs interleaved (gemm_f32.hip, PM form)
  float *l0f_Wt_blk = nullptr;                      // l0f_Wt in 256-column blocks (layer 0 wider than 256 columns)
  float *hd_W0fp = nullptr;                         // hd_W0t column-permuted for the fused exact-f32 kernel
  float *l0af_W = nullptr, *l0af_shift = nullptr;   // layer 0 aggregate-first (bf16 path): per-head bf16 images of l0f_Wt; layer 0's shift + scale * l0f_b
  std::vector<BgnnLayer> layers;
  float *ones = nullptr;      // [256] of 1.0f: the identity scale of an unfolded epilogue
  int head_hidden_total;      // (2 or 3) * hid/2, padded to a multiple of 32
  float *hd_W0t, *hd_b0;      // [hid][head_hidden_total], [head_hidden_total]
  float *hd_W0sp = nullptr, *hd_W0sp16 = nullptr;   // hd_W0t as bf16 / float16 hi / lo split images
  float hd_W0sp16_inv = 1.0f, l0f_Wsp16_inv = 1.0f; // 2^-S of the float16 images (pack_split)
  float *hd_W1, *hd_b1;       // second layers packed: cls [classes][hid/2], conf [hid/2], corr [hid/2]; biases
  float *hd_tab = nullptr;    // fused heads epilogue's LDS image, [296]: b0 [96] | second-layer rows [<= 6][32] | their biases [8]
  // The fused layer kernels rebuild a slot's attributes in the CANONICAL order (distance, depth_difference, slope).  For a graph built
  // with another edge feature list (any selection / order / repetition of those three and "zero"), the folded edge vector
  // V [heads][edge_dim] of every layer is re-expressed over the canonical three: V3[h][id] = sum of V[h][j] over the list positions j
  // that hold attribute id.  One small device table [layers][heads][3] per list, made on first use (model_canonical_V).
  std::vector<float> h_V;                             // host copy of every layer's V: [layers][heads_l][edge_dim]
  std::vector<std::pair<uint32_t, float *>> v3_tables;   // (packed list, device table [layers][max heads][3])
};

struct bgnn_graph {
  bgnn_ctx *ctx;
  int kind;                   // 0 = stencil grid graph (ELL), 1 = generic CSR
  int32_t n_tiles = 0;
  int32_t total_cells = 0;    // grid: number of cells; generic: number of nodes
  int32_t K = 0;              // ELL width (stencil size)
  int32_t F = 0;              // node features
  int32_t ED = 0;             // edge features
  int32_t include_self_loops = 0;
  int32_t has_unc = 0;
  std::vector<BgnnTileMeta> h_tiles;
  // device
  BgnnTileMeta *d_tiles = nullptr;
  BgnnWorkItem *d_items = nullptr;
  int32_t n_items = 0;
  int32_t *d_node_id = nullptr;       // [cells]  >=0 node id ; <0 : -(prefix+1)
  int32_t *d_cell_of_node = nullptr;  // [cells]
  int64_t *d_counts = nullptr;        // [0]=n_nodes [1]=n_edges(valid after export scan)
  int64_t *d_n_nodes_copy = nullptr;  // caller's copy of the node count (bgnn_infer_tiles: written by the compaction scan itself)
  bool tables_cached = false;         // d_tiles / d_items belong to the context's table cache (uniform batches)
  uint64_t table_cache_id = 0;        // ... the entry this graph holds a reference on (0: none)
  float *d_x8 = nullptr;              // [rows][8]
  float *d_local_std = nullptr;       // [rows]
  int32_t *d_nbr = nullptr;           // grid: [rows][K] ; generic: col[E]
  // grid graphs: the stencil id table is built ON DEMAND (ensure_stencil_table): the fused layer kernels never read it -- only
  // the export, the edge counts, the generic aggregate and the non-attention backbones do -- and writing it was 32 (k = 8) / 64
  // (k = 16) of the feature kernel's 188 / 380 bytes per node
  mutable bool nbr_valid = false;
  // grid graphs with the default edge feature list are built COMPACT (graph_build.hip, FeatureArgs): slopes, node depths and the
  // tiles' edge lengths; the fused layer kernels read those, d_eattr is expanded on demand (ensure_edge_attrs)
  bool compact_edges = false;
  int32_t edge_ids[4] = {0, 1, 2, 3};  // the edge feature list the graph was built with (BGNN_EF_*), ED entries
  bool edge_default = false;          // ... is [distance, depth_difference, slope]
  mutable bool eattr_valid = false;
  float *d_slope = nullptr;           // [rows][K]
  float *d_node_depth = nullptr;      // [rows]
  float4 *d_tile_dist = nullptr;      // [n_tiles]
  int32_t *d_atlas_tile_of = nullptr; // canvas: grid index of every canvas cell that holds a node (with d_atlas; other cells: undefined)
  float *clear_grids[3] = {nullptr, nullptr, nullptr};   // bgnn_infer_tiles: result grids the canvas fill zero-fills on its way
  bool grids_cleared = false;
  mutable float *d_eattr = nullptr;   // grid: [rows][K][ED] ; generic: [E][ED]
  int32_t *d_rowptr = nullptr;        // generic only [N+1]
  int32_t *d_edge_perm = nullptr;     // generic only
  int64_t n_nodes_host = -1;          // cached after a sync
  int64_t n_edges_host = -1;
  int64_t generic_E = 0;
  int32_t row_capacity = 0;           // rows allocated for node-indexed arrays
  // 2-D cell blocks (16x16) for the LDS-tiled aggregate
  BgnnWorkItem *d_items2 = nullptr;   // {tile, r0, c0}; unused when every tile has the same shape
  int32_t n_blocks2 = 0;
  int32_t uni_h = 0, uni_w = 0;       // common tile shape, 0 if ragged
  int32_t bh2 = 0, bw2 = 0;           // blocks per tile (uniform case)
  // 8x16 cell blocks for the fused layer kernel
  BgnnWorkItem *d_items3 = nullptr;
  int32_t n_blocks3 = 0, bh3 = 0, bw3 = 0;
  int32_t max_w = 0;
  // Atlas of a RAGGED batch (refinement grids of 3..50 cells a side): the grids are shelf-packed, with a gutter of invalid cells
  // as wide as the stencil's reach, into one canvas whose 8x16 blocks the fused layer kernels walk instead of per-grid blocks
  // (which are 68 % full on config 4's grids; the canvas is ~85 % full).  d_atlas[canvas cell] = node id or -1.
  int32_t *d_atlas = nullptr;
  char *d_tables = nullptr;               // ragged batch: ONE block with every host-built table (tiles, items*, canvas tables)
  BgnnTileMeta *d_atlas_tile = nullptr;   // the canvas as ONE tile {h, w, cell_off = 0}
  int32_t *d_atlas_pos = nullptr;         // [n_tiles][2] = (row, column) of each grid's origin on the canvas
  int32_t atlas_h = 0, atlas_w = 0;
};

namespace bgnn {

int ctx_workspace(bgnn_ctx *ctx, int slot, size_t bytes, void **out);
int ctx_upload(bgnn_ctx *ctx, const void *host, size_t bytes, void *dev);

// profiling scope: records events around a kernel launch when enabled for that kernel
struct ProfScope {
  bgnn_ctx *ctx;
  int idx;
  ProfScope(bgnn_ctx *c, int kernel);
  ~ProfScope();
};

// ---- launchers implemented in the kernel translation units ------------------------------
int launch_graph_build(bgnn_ctx *ctx, bgnn_graph *g, const bgnn_tiles *tiles, const bgnn_graph_opts *opts);
int launch_graph_export(bgnn_graph *g, float *x, int64_t *edge_index, float *edge_attr, float *pos,
                        int64_t *valid_rows, int64_t *valid_cols, float *local_std, int64_t *batch);
int ensure_stencil_table(const bgnn_graph *g);
    // builds d_nbr of a grid graph if it has not been built yet (stream-ordered)
int ensure_edge_attrs(const bgnn_graph *g);      // full edge-attribute table of a compact graph, on demand (stream-ordered)
int launch_graph_count_edges(bgnn_graph *g);
int launch_graph_scatter(bgnn_graph *g, const float *node_values, float fill, float *grid);
int launch_results_to_grids(bgnn_graph *g, const int64_t *cls, const float *conf, const float *corr,
                            float norm_floor, float *cls_grid, float *conf_grid, float *corr_grid);
int launch_generic_build(bgnn_ctx *ctx, bgnn_graph *g, int64_t n_nodes, int32_t n_feat, const float *x,
                         int64_t n_edges, const int64_t *edge_index, int32_t edge_dim, const float *edge_attr);

// Y[M,NC] = act(X[M,K] @ Wt[K,NC] + bias); M read from d_counts[0] (bounded by max_rows)
// optional fused epilogue: asd[M][2H] = (Y . att_src, Y . att_dst) per head (att_src == nullptr: off)
int launch_gemm_f32(bgnn_ctx *ctx, const float *X, int ldx, const float *Wt, const float *bias, float *Y,
                    int ldy, const int64_t *d_m, int64_t max_rows, int K, int NC, int relu,
                    const float *att_src = nullptr, const float *att_dst = nullptr, float *asd = nullptr,
                    int H = 0, int C = 0, const float *Wt_split = nullptr, int split_mode = 0,
                    const float *front_W0t = nullptr, const float *front_b0 = nullptr, const float *Wt_pm = nullptr,
                    const float *Wt_blk = nullptr, float split_inv_scale = 1.0f);
bool gemm_front_available(const bgnn_ctx *ctx, int64_t max_rows, int NC, int split_mode);
// bf16 path, layer 0 aggregate-first: extractor layer 1 -> h1 [M][64] bf16 + layer 0's attention dots through the bf16 front GEMM's alpha tile
int launch_extractor_af(bgnn_ctx *ctx, const float *x8, const float *W0t, const float *b0, const float *alpha_tile, void *h1, float *asd,
                        const int64_t *d_m, int64_t max_rows, int H);
// training-mode dropout (bgnn.h, bgnn_dropout): one counter-based draw per element, see there
struct DropSpec {
  uint32_t thr = 0;          // keep <=> hash >= thr   (floor(p * 2^32); 0: nothing is dropped)
  float scale = 1.0f;        // 1 / (1 - p)
  uint64_t seed = 0;
  uint32_t stream = 0;
};
__host__ __device__ inline uint32_t bgnn_drop_hash(uint64_t seed, uint32_t stream, uint64_t index) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)stream + 1ull) + 0xD1B54A32D192ED03ull * index;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 32);
}
inline DropSpec make_drop_spec(float p, uint64_t seed, uint32_t stream) {
  DropSpec d;
  if (p > 0.0f) {
    const double t = (double)p * 4294967296.0;
    d.thr = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
    d.scale = (float)(1.0 / (1.0 - (double)p));
  }
  d.seed = seed; d.stream = stream;
  return d;
}
// x [M][width] (leading dimension ld) *= keep / (1 - p), in place; M from d_m
int launch_dropout(bgnn_ctx *ctx, float *x, int width, int ld, const int64_t *d_m, int64_t max_rows, const DropSpec &d);
int launch_gat_aggregate(bgnn_ctx *ctx, const bgnn_graph *g, const BgnnLayer &L, int C, int ED, const float *xw,
                         const float *asd, float *out, int relu, const DropSpec *attention_drop = nullptr);
int launch_gat_aggregate_tiled(bgnn_ctx *ctx, const bgnn_graph *g, const BgnnLayer &L, int C, int ED, const float *xw,
                               const float *asd, float *out, int relu);
// fused K4 + next K3 (EPI_NEXT) / K4(last) + K5 + K6 (EPI_HEADS); BGNN_ERR_UNSUPPORTED when no instance fits
// (V3: the layer's edge vector over the canonical three attributes, [heads][3] -- nullptr: L.V is that already, default list)
int launch_fused_layer_next(bgnn_ctx *ctx, const bgnn_graph *g, const BgnnLayer &L, const BgnnLayer &Ln, int C, const float *V3,
                            const void *xw, const float *asd, void *xw_next, float *asd_next);
// ... layer 0 of the bf16 path from the extractor's h1 (aggregate, then the folded lin_0 weight per head, then lin_1): gat_layer_fused.hip
int launch_fused_layer0_af(bgnn_ctx *ctx, const bgnn_graph *g, const BgnnLayer &L, const BgnnLayer &Ln, int C, const float *V3,
                           const void *h1, const float *asd, const float *W0af, const float *shift_af, void *xw_next, float *asd_next);
bool fused_heads_available(const bgnn_ctx *ctx, const bgnn_graph *g, const bgnn_model *m);
// one layer of a plain backbone as aggregate -> GEMM -> post-op (mode 1 GCN, 2 GraphSAGE, 3 GIN's first Linear); see gat_layer_fused.hip
int launch_fused_plain_layer(bgnn_ctx *ctx, const bgnn_graph *g, int mode, int C, const float *x, const float *dinv, const float *Wfp,
                             const float *ones, const float *post_scale, const float *post_shift, int post_relu, float *out);
int launch_fused_layer_heads(bgnn_ctx *ctx, const bgnn_graph *g, const bgnn_model *m, const BgnnLayer &L, int C, const float *V3,
                             const void *xw, const float *asd, float thr_auto, float thr_review, float norm_floor,
                             const bgnn_outputs *o, float *cls_grid, float *conf_grid, float *corr_grid);
int launch_degree_inv_sqrt(bgnn_ctx *ctx, const bgnn_graph *g, float *dinv);
int launch_neighbor_reduce(bgnn_ctx *ctx, const bgnn_graph *g, int mode, const float *x, int D, const float *dinv,
                           const float *scale, const float *shift, int relu, float *out, int ldo, float *copy_self);
size_t bn_train_workspace_bytes(int W);
int launch_bn_train(bgnn_ctx *ctx, float *z, int ld, int W, int64_t max_rows, const int64_t *d_m, const float *bn_w,
                    const float *bn_b, float eps, int relu, void *workspace, float *batch_mean, float *batch_var_unbiased);
int launch_heads_final(bgnn_ctx *ctx, const bgnn_model *m, const float *hid, int ldh, const int64_t *d_m,
                       int64_t max_rows, float thr_auto, float thr_review, const bgnn_outputs *o);

}  // namespace bgnn
