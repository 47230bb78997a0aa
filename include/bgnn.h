/*
 * bgnn.h -- C ABI of libbgnn_hip.so: the MI355X-native hot path of Bathymetric-GNN.
 *
 * The reference (grant-froelich/Bathymetric-GNN) is pure Python and has no FFI of its
 * own; the drop-in boundary is its Python class API.  Each entry point below names the
 * reference interface it replaces (file:line into the reference tree).  The Python
 * mirror of that API (bathymetric-gnn_amd/{data,models,scripts}) binds these symbols
 * with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C types only; every pointer marked DEVICE is a HIP device pointer on the
 *     context's GPU (a torch tensor's data_ptr()); HOST pointers are ordinary memory.
 *   - all calls return 0 (BGNN_OK) or a negative error code; bgnn_last_error() gives the
 *     thread-local message of the last failure.
 *   - all work is enqueued on the context's HIP stream and is asynchronous unless the
 *     function is documented to synchronise.  A context is not re-entrant; different
 *     contexts (one per GPU / per worker) are independent.
 *   - caller owns every buffer it passes in; the library owns handles it returns and the
 *     context-scoped arenas behind them (freed by the *_destroy calls).
 */
#ifndef BGNN_H
#define BGNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BGNN_ABI_VERSION 6

#define BGNN_OK 0
#define BGNN_ERR_INVALID (-1)     /* bad argument (-> ValueError in the Python mirror)   */
#define BGNN_ERR_HIP (-2)         /* HIP runtime failure (-> RuntimeError)               */
#define BGNN_ERR_NOMEM (-3)       /* device allocation failed                            */
#define BGNN_ERR_UNSUPPORTED (-4) /* valid in the reference, not built here              */

typedef struct bgnn_ctx bgnn_ctx;
typedef struct bgnn_model bgnn_model;
typedef struct bgnn_graph bgnn_graph;

int bgnn_abi_version(void);
const char *bgnn_last_error(void);
/* Identity of the kernels inside this library: the first 16 hex digits of the sha256 over its kernel sources (the .hip and .h files under csrc/,
 * sorted by name), stamped at build time ("unknown" for a build outside __graft_entry__.build()).  Measurement tooling uses it to
 * refuse counter data collected on other kernels (bench.py: roofline.traffic); no reference counterpart. */
const char *bgnn_build_id(void);

/* ---- context: one per GPU worker --------------------------------------------------- */
/* stream == NULL: the context creates and owns a non-blocking HIP stream; otherwise it
 * enqueues on the caller's hipStream_t (e.g. torch.cuda.Stream().cuda_stream).
 * Replaces: the device selection in models/pipeline.py:73-88 and
 * scripts/inference_native.py:381-394. */
int bgnn_ctx_create(int device, void *stream, bgnn_ctx **out);
/* Synchronises the stream and releases the context's arenas AND every graph built on it that has not been destroyed yet:
 * their bgnn_graph handles are invalid afterwards and must not be passed to any call, bgnn_graph_destroy included.
 * Models are separate allocations: destroy them first. */
int bgnn_ctx_destroy(bgnn_ctx *ctx);
int bgnn_ctx_synchronize(bgnn_ctx *ctx);
void *bgnn_ctx_stream(bgnn_ctx *ctx);

/* Run-time switches of a context (all int-valued).  Defaults are taken from the environment ONCE, when the context is
 * created (the variable in brackets); afterwards only this call changes them:
 *   "matrix_path"     0 exact f32 (default), 1 bf16x3, 2 fp16x3: opt-in operand-split MFMA paths [BGNN_SPLIT_BF16 / BGNN_SPLIT_F16]
 *                     (hi + lo parts of both operands, three 16-bit MFMAs, float32 accumulate; fp16x3 keeps its weight images scaled
 *                     into float16's normal range and lands as close to a float64 forward as the exact path, while |activations| < 65504);
 *                     3 bf16: layer activations stored as bf16 in HBM and multiplied on the bf16 MFMA, float32 softmax /
 *                     aggregation / accumulation (BASELINE config 3 "bf16 node features"; no 1e-4 contract) [BGNN_BF16]
 *   "fused"           1 (default): K4 fused with the next K3 / K5 + K6; 0: separate kernels           [BGNN_NO_FUSED]
 *   "fold_extractor"  1 (default): extractor layer 2 folded into lin of GAT layer 0; 0: unfolded chain [BGNN_NO_FOLD]
 *   "fused_front"     1 (default): feature extractor layer 1 runs inside the lin_0 GEMM where that GEMM takes its W-resident form
 *                     (same instructions, bit-identical, one launch and 512 B/node of traffic less); 0: own launch [BGNN_NO_FUSED_FRONT]
 *   "fused_persistent" 0 (default) / 1: opt-in experiment -- big uniform batches on the exact path run the 256 -> 256 fused layer as
 *                     one persistent workgroup per CU (bit-identical results, currently slower)            [BGNN_PERSISTENT]
 *   "ragged_atlas"    1 (default): for ragged batches the fused layers walk a shelf-packed canvas of the grids (denser 8x16
 *                     blocks); 0: per-grid blocks                                                      [BGNN_NO_ATLAS]
 *   "features_tiled"  1 (default): node features / stencil table / edge attributes by the LDS-tiled kernel that computes the
 *                     float64 slope of an edge once for both of its directions (K = 8 / 16, 3 edge features) for uniform batches
 *                     of tiles at least 64 cells wide; 0: always the thread-per-cell kernel (bit-identical results; the form the
 *                     tiled one is tested against); 2: the tiled kernel for every shape (tests)
 *   "gemm_pair_major" 1 (default): the exact-f32 lin_0 GEMM with the extractor in front runs its MFMAs tile-pair-major, the
 *                     epilogue of pair p issued between the MFMAs of pair p + 1; 0: tile-major, epilogue after (bit-identical;
 *                     the form it is tested against)                                              [BGNN_NO_PAIR_MAJOR]
 *   "bf16_two_phase"  1 (default): matrix_path 3 runs its 256 -> 256 fused layer in two phases (aggregate to bf16 registers, then
 *                     the GEMM in column passes; bit-identical to 0, the one-phase instance)          [BGNN_NO_TWO_PHASE]
 *   "bf16_layer0_af"  1 (default): matrix_path 3, default model shape: layer 0 aggregates the extractor's 64-channel output and applies the
 *                     folded lin_0 weight per head afterwards, inside the fused launch (GATConv's sum is linear; no lin_0 GEMM launch, no
 *                     256-channel lin_0 rows in HBM; needs bf16_two_phase = 1); 0: lin_0 GEMM first (another rounding sequence)   [BGNN_NO_LAYER0_AF]
 *   "stats_narrow"    -1 (default): 16-wide box-statistics workgroups when the 64-wide launch would leave CUs idle; 0 / 1 force
 * plus experiment / diagnostic knobs ("fused_lds_pad_kb", "gemm_waves", "gemm_no_wres"; "diag_mask",
 * "diag_stamps", "gemm_diag" exist only in the diagnostic build of the library).  Unknown names -> BGNN_ERR_INVALID. */
int bgnn_ctx_set_option(bgnn_ctx *ctx, const char *name, int value);
int bgnn_ctx_get_option(bgnn_ctx *ctx, const char *name, int *value);

/* Per-kernel device timing with HIP events on the context's stream.  kernel_mask selects
 * kernels by (1u << BGNN_K_*); 0 switches profiling off.  bgnn_ctx_profile_read
 * synchronises, adds the elapsed time of every recorded launch to ms[BGNN_K_COUNT] /
 * launches[BGNN_K_COUNT] (HOST arrays) and clears the record list. */
enum {
  BGNN_K_SCAN = 0,      /* valid-cell compaction scans                                  */
  BGNN_K_STATS = 1,     /* K1a masked 5x5 box statistics (fp64)                         */
  BGNN_K_FEATURES = 2,  /* K1b/K2 node features + stencil neighbour table + edge attrs  */
  BGNN_K_EXPORT = 3,    /* PyG-layout materialisation (edge_index int64, ...)           */
  BGNN_K_GEMM = 4,      /* K3 dense node-feature x weight GEMM (f32 MFMA)               */
  BGNN_K_ATTCOEF = 5,   /* alpha_src / alpha_dst dot products (fused into K3's epilogue) */
  BGNN_K_AGGREGATE = 6, /* K4 gather - per-node softmax - weighted aggregate (+BN+ReLU) */
  BGNN_K_HEADS = 7,     /* K5 output heads                                              */
  BGNN_K_SCATTER = 8,   /* K6 node -> grid scatter + correction de-normalisation        */
  BGNN_K_FUSED = 9,     /* K4 fused with the next layer's K3 (or with K5 + K6 for the last) */
  BGNN_K_COUNT = 10
};
int bgnn_ctx_profile(bgnn_ctx *ctx, uint32_t kernel_mask);
int bgnn_ctx_profile_read(bgnn_ctx *ctx, double *ms, int64_t *launches);

/* ---- model: BathymetricGNN (models/gnn.py:263-358) ------------------------------------ */
enum { /* GNNBackbone gnn_type (models/gnn.py:120-143); GAT is the hot path, the others run on plain gather + GEMM kernels */
  BGNN_GNN_GAT = 0, BGNN_GNN_GCN = 1, BGNN_GNN_SAGE = 2, BGNN_GNN_GIN = 3
};
typedef struct bgnn_model_desc {
  int32_t in_channels;        /* 7, or 8 with the uncertainty column                     */
  int32_t hidden;             /* 64 (the fused kernels' width); 32 and 128 run generic kernels */
  int32_t num_layers;         /* num_gnn_layers (>= 1)                                   */
  int32_t heads;              /* 4; GAT only, power of two, heads * hidden <= 512 (above 256: generic kernels, two column blocks); last layer: 1 head, mean (gnn.py:125-132) */
  int32_t num_classes;        /* 3                                                       */
  int32_t edge_dim;           /* 3                                                       */
  int32_t predict_correction; /* correction head present                                 */
  float bn_eps;               /* torch BatchNorm1d eps, 1e-5                             */
  int32_t gnn_type;           /* BGNN_GNN_*; 0 = GAT                                     */
} bgnn_model_desc;

/* Number of float32 values bgnn_model_create expects in `weights`, in this order
 * (torch / torch_geometric state_dict names, each tensor row-major as torch stores it):
 *   feature_extractor.mlp.0.{weight[hid,in],bias[hid]}, feature_extractor.mlp.3.{weight[hid,hid],bias[hid]}
 *   for l in 0..L-1 (H_l = heads, or 1 for the last layer; D_l = hid for l=0 else hid*heads;
 *                    W_l = H_l*hid, or hid for the last layer):
 *     gnn.convs.l.lin.weight[H_l*hid, D_l], att_src[H_l*hid], att_dst[H_l*hid], att_edge[H_l*hid],
 *     gnn.convs.l.lin_edge.weight[H_l*hid, edge_dim], gnn.convs.l.bias[W_l],
 *     gnn.norms.l.module.{weight, bias, running_mean, running_var}[W_l]
 *   classification_head.mlp.0.{weight[hid/2,hid],bias}, classification_head.mlp.3.{weight[classes,hid/2],bias}
 *   confidence_head.mlp.0.{..}, confidence_head.mlp.3.{weight[1,hid/2],bias[1]}
 *   correction_head.mlp.0.{..}, correction_head.mlp.3.{..}     (only if predict_correction)
 * For the other backbones the per-layer block is (every layer hid -> hid, torch_geometric default arguments):
 *   GCN : gnn.convs.l.lin.weight[hid,hid], gnn.convs.l.bias[hid], norms (4 x [hid])
 *   SAGE: gnn.convs.l.lin_l.weight[hid,hid], lin_l.bias[hid], lin_r.weight[hid,hid], norms
 *   GIN : gnn.convs.l.nn.0.weight[hid,hid], nn.0.bias[hid], nn.2.weight[hid,hid], nn.2.bias[hid], norms
 */
size_t bgnn_model_weight_count(const bgnn_model_desc *desc);

/* Replaces BathymetricGNN.__init__ + load_state_dict + .to(device).eval()
 * (models/pipeline.py:115-130, scripts/inference_native.py:406-419).  `weights` is HOST
 * memory; the library folds lin_edge.att_edge -> V[H,edge_dim] and BatchNorm(+conv bias)
 * -> scale/shift in float64 and uploads the packed result.  Synchronises. */
int bgnn_model_create(bgnn_ctx *ctx, const bgnn_model_desc *desc, const float *weights,
                      size_t n_weights, bgnn_model **out);
int bgnn_model_destroy(bgnn_model *model);

/* ---- graph construction: GraphBuilder.build_graph (data/graph_construction.py:91-174) */
enum { /* node feature ids, graph_construction.py:288-306 */
  BGNN_NF_DEPTH = 0, BGNN_NF_LOCAL_MEAN = 1, BGNN_NF_LOCAL_STD = 2, BGNN_NF_GRADIENT_X = 3,
  BGNN_NF_GRADIENT_Y = 4, BGNN_NF_GRADIENT_MAGNITUDE = 5, BGNN_NF_CURVATURE = 6,
  BGNN_NF_UNCERTAINTY = 7
};
enum { /* edge feature ids, graph_construction.py:346-367 (anything else is 0.0) */
  BGNN_EF_DISTANCE = 0, BGNN_EF_DEPTH_DIFFERENCE = 1, BGNN_EF_SLOPE = 2, BGNN_EF_ZERO = 3
};

typedef struct bgnn_tiles { /* a batch of independent grids (tiles / refinement grids) */
  int32_t n_tiles;
  const int32_t *hw;         /* HOST [n_tiles][2] = (height, width), each >= 2           */
  const double *resolution;  /* HOST [n_tiles][2] = (res_x, res_y)                       */
  const float *depth;        /* DEVICE, tiles concatenated, each row-major [h][w]        */
  const uint8_t *mask;       /* DEVICE, 1 = valid cell (GraphBuilder's valid_mask)       */
  const float *uncertainty;  /* DEVICE or NULL                                           */
} bgnn_tiles;

typedef struct bgnn_graph_opts {
  int32_t connectivity;       /* 4 or 8 (graph_construction.py:78-89); 16 = dilated ext. */
  int32_t include_self_loops; /* appended last in the exported edge list (:226-229)      */
  int32_t n_node_features;    /* <= 8, ids above in output column order; uncertainty is  */
  int32_t node_features[8];   /*   appended by the library when given and not listed     */
  int32_t n_edge_features;    /* 1..4                                                    */
  int32_t edge_features[4];
} bgnn_graph_opts;

/* Builds the device-resident graph for every tile of the batch: row-major valid-cell
 * compaction, stencil neighbour table, node features, edge attributes.  Asynchronous. */
int bgnn_graph_build(bgnn_ctx *ctx, const bgnn_tiles *tiles, const bgnn_graph_opts *opts,
                     bgnn_graph **out);

/* A graph given as PyG tensors (a `Data` built elsewhere): x[N,F], edge_index[2,E] int64
 * (row 0 = source, row 1 = target), edge_attr[E,edge_dim]; all DEVICE.  Replaces the
 * (x, edge_index, edge_attr) triple read at models/gnn.py:381-383.  Synchronises. */
int bgnn_graph_from_edges(bgnn_ctx *ctx, int64_t n_nodes, int32_t n_feat, const float *x,
                          int64_t n_edges, const int64_t *edge_index, int32_t edge_dim,
                          const float *edge_attr, bgnn_graph **out);
int bgnn_graph_destroy(bgnn_graph *graph);

/* Sizes (synchronises).  node_off / edge_off: HOST [n_tiles+1] prefix offsets of each
 * tile's nodes / edges in the batch (PyG Batch.from_data_list order); may be NULL. */
int bgnn_graph_counts(bgnn_graph *graph, int64_t *n_nodes, int64_t *n_edges, int32_t *n_feat,
                      int32_t *edge_dim, int64_t *node_off, int64_t *edge_off);

/* Materialise the PyG-compatible tensors of the Data object (graph_construction.py:144-167,
 * batched as torch_geometric Batch.from_data_list does): any pointer may be NULL.
 * All DEVICE: x[N,F] f32, edge_index[2,E] i64 (offset-major, node-ascending per tile;
 * bit-exact with the reference), edge_attr[E,edge_dim] f32, pos[N,2] f32 = (col,row),
 * valid_rows/valid_cols[N] i64, local_std[N] f32, batch[N] i64. */
int bgnn_graph_export(bgnn_graph *graph, float *x, int64_t *edge_index, float *edge_attr,
                      float *pos, int64_t *valid_rows, int64_t *valid_cols, float *local_std,
                      int64_t *batch);

/* GraphBuilder.graph_to_grid (graph_construction.py:471-505) for every tile of the batch:
 * grid (DEVICE, same layout as bgnn_tiles.depth) = fill, then grid[valid cell] = value. */
int bgnn_graph_scatter(bgnn_graph *graph, const float *node_values, float fill, float *grid);

/* ---- forward: BathymetricGNN.forward / .predict (models/gnn.py:360-451) ------------- */
typedef struct bgnn_outputs { /* all DEVICE, any may be NULL */
  float *class_logits;      /* [N, classes]                                              */
  float *class_probs;       /* [N, classes]                                              */
  int64_t *predicted_class; /* [N]                                                       */
  float *confidence;        /* [N]                                                       */
  float *correction;        /* [N] (normalised units)                                    */
  int64_t *action;          /* [N] 0 keep / 1 auto-correct / 2 review (gnn.py:436-445)   */
  uint8_t *needs_review;    /* [N]                                                       */
  uint8_t *auto_correct;    /* [N]                                                       */
  float *hidden;            /* [N, hidden] backbone output (diagnostics / parity tests)  */
} bgnn_outputs;

int bgnn_forward(bgnn_ctx *ctx, bgnn_model *model, bgnn_graph *graph, float auto_correct_threshold,
                 float review_threshold, const bgnn_outputs *out);

/* The model's plain-torch sub-modules on their own, so that they can be held to vectors the reference itself produced
 * (tests/golden/model_*.npz): LocalFeatureExtractor.forward (models/gnn.py:34-71; called at :386) on x [n_nodes][in_channels]
 * -> out [n_nodes][hidden] (the unfolded two-Linear chain), and the three heads + softmax / argmax / sigmoid + predict's
 * flags (models/gnn.py:191-260, :392-406, :427-449) on a backbone output hidden [n_nodes][hidden].  All DEVICE, row-major
 * float32; `out->hidden` must be NULL for bgnn_heads.  Asynchronous. */
int bgnn_feature_extractor(bgnn_ctx *ctx, bgnn_model *model, const float *x, int64_t n_nodes, float *out);
int bgnn_heads(bgnn_ctx *ctx, bgnn_model *model, const float *hidden, int64_t n_nodes, float auto_correct_threshold,
               float review_threshold, const bgnn_outputs *out);

/* BathymetricGNN.forward with the module in train() mode and every dropout probability 0 (models/gnn.py:360-408 with
 * :151-154, :179-186 in training mode): each BatchNorm layer normalises with the mean and the biased variance of THIS
 * batch of nodes instead of its running statistics (torch.nn.BatchNorm1d, which torch_geometric's BatchNorm wraps).
 * Forward only -- there is no backward pass in this library.  Dropout with p > 0 draws from torch's generator and is
 * not reproduced: the host layer refuses that case.
 *   bn_batch_mean, bn_batch_var  DEVICE f32 [sum over layers of the layer width], layer after layer; either may be
 *       NULL.  bn_batch_var is the UNBIASED variance: what the caller blends into running_var
 *       (running = (1 - momentum) * running + momentum * batch), as running_mean with bn_batch_mean.
 *   out: as for bgnn_forward; action / needs_review / auto_correct must be NULL (they belong to predict()).
 * A batch of exactly one node is refused like torch does ("Expected more than 1 value per channel when training").
 * Synchronises the stream once (reads the node count). */
int bgnn_forward_train(bgnn_ctx *ctx, bgnn_model *model, bgnn_graph *graph, float *bn_batch_mean, float *bn_batch_var,
                       const bgnn_outputs *out);

/* The same forward with ACTIVE dropout (ABI 6).  The reference drops at four places, all reproduced here:
 *   p_extractor  nn.Dropout after the ReLU of LocalFeatureExtractor's first Linear (models/gnn.py:55-57)
 *   p_attention  GATConv(dropout=...): on the attention coefficients after the softmax, self loops included (:125-132)
 *   p_features   F.dropout after the ReLU of every GNN layer but the last (:184-186)
 *   p_heads      nn.Dropout on the hidden units of the three heads (:206, :229, :253)
 * A kept value is multiplied by 1 / (1 - p) (float32), a dropped one is 0.  WHICH values are dropped cannot follow torch's
 * generator (its stream differs between CPU and GPU builds of torch itself); it is a documented counter-based Bernoulli
 * draw instead, a pure function of (seed, place, element) -- reproducible, independent of launch geometry, and restated on
 * the host by oracle/gat_cpu.py so that the parity tests run the oracle with the very same masks:
 *   z = seed + 0x9E3779B97F4A7C15 * (stream + 1) + 0xD1B54A32D192ED03 * index        (mod 2^64)
 *   z = (z ^ z >> 30) * 0xBF58476D1CE4E5B9;  z = (z ^ z >> 27) * 0x94D049BB133111EB;  z ^= z >> 31      (splitmix64's finaliser)
 *   keep  <=>  (z >> 32) >= floor(p * 2^32)
 * stream / index: extractor 1 / row * hidden + column; heads 2 / row * (number of heads * hidden / 2) + column (classification |
 * confidence | correction units side by side); features 64 + layer / row * width + column; attention 16 + layer /
 * ((target << 32 | source) * heads + head), a self loop having source == target (parallel edges of a foreign graph therefore
 * share one draw).  Every p must lie in [0, 1). */
typedef struct bgnn_dropout {
  float p_extractor, p_attention, p_features, p_heads;
  uint64_t seed;
} bgnn_dropout;
int bgnn_forward_train_dropout(bgnn_ctx *ctx, bgnn_model *model, bgnn_graph *graph, const bgnn_dropout *dropout,
                               float *bn_batch_mean, float *bn_batch_var, const bgnn_outputs *out);

/* BathymetricPipeline._process_tile (models/pipeline.py:243-314) and
 * NativeVRProcessor._extract_results_from_outputs (scripts/inference_native.py:181-204)
 * for a whole batch, fused: build -> predict -> scatter.  Output grids (DEVICE, layout of
 * bgnn_tiles.depth, fill 0.0): classification (class id as float), confidence,
 * correction = normalised correction * max(local_std, norm_floor).  Asynchronous.
 * n_nodes_out: optional DEVICE int64[1] receiving the number of classified nodes. */
int bgnn_infer_tiles(bgnn_ctx *ctx, bgnn_model *model, const bgnn_tiles *tiles,
                     const bgnn_graph_opts *opts, float auto_correct_threshold, float review_threshold,
                     float norm_floor, float *classification, float *confidence, float *correction,
                     int64_t *n_nodes_out);

/* ---- stitching: TileMerger + BathymetricPipeline.process post-steps on the device ------------------
 * Replaces, for one survey grid [height][width]: TileMerger.add_tile / finalize (data/tiling.py:384-454:
 * Hann-ramp weighted blend of confidence and correction, confidence-arbitrated classification), the
 * preservation of valid cells no processed tile covered (models/pipeline.py:196-207: class 0, confidence 0,
 * correction 0) and _apply_corrections (models/pipeline.py:316-349).  Tiles are visited per cell in
 * ascending spec order, so results equal the reference's serial host merge bit for bit.
 *   row_start/row_end [n_tile_rows], col_start/col_end [n_tile_cols]   DEVICE int32 (TileSpec extents)
 *   row_weights [n_tile_rows][weight_pitch], col_weights [n_tile_cols][weight_pitch]  DEVICE f32: the 1-D
 *       blend windows of each extent as TileManager._create_1d_blend computes them (host numpy)
 *   tile_offsets [n_tile_rows*n_tile_cols] DEVICE int64: offset of the tile's cells inside classification /
 *       confidence / correction (the concatenated outputs of bgnn_infer_tiles); negative = tile was skipped
 *   outputs [height][width] DEVICE f32; cells no tile covered and that are not valid stay NaN.   Asynchronous. */
int bgnn_stitch_tiles(bgnn_ctx *ctx, int32_t height, int32_t width, int32_t n_tile_rows, int32_t n_tile_cols,
                      const int32_t *row_start, const int32_t *row_end, const int32_t *col_start,
                      const int32_t *col_end, const float *row_weights, const float *col_weights,
                      int32_t weight_pitch, const int64_t *tile_offsets, const float *classification,
                      const float *confidence, const float *correction, const float *depth,
                      const uint8_t *valid_mask, float auto_correct_threshold, float *out_classification,
                      float *out_confidence, float *out_correction, float *out_cleaned_depth);

/* Tiles out of a survey resident in HBM: TileManager.extract_tile / iterate_tiles (data/tiling.py:136-216) for a
 * batch of n_tiles equally sized windows (the shift-back rule, :119-122, makes every tile of a survey the same
 * size).  origins DEVICE int32 [n_tiles][2] = (row_start, col_start); the caller guarantees origin + tile size
 * stays inside the survey.  bgnn_cut_tiles copies depth / valid mask / uncertainty (NULL pair allowed) windows into
 * the concatenated tile layout of bgnn_tiles; bgnn_tile_valid_counts gives the per-tile number of valid cells
 * (DEVICE int64 [n_tiles]) that iterate_tiles' min_valid_ratio test needs.  Asynchronous. */
int bgnn_cut_tiles(bgnn_ctx *ctx, int32_t height, int32_t width, const float *depth, const uint8_t *valid_mask,
                   const float *uncertainty, int32_t n_tiles, const int32_t *origins, int32_t tile_h, int32_t tile_w,
                   float *out_depth, uint8_t *out_mask, float *out_uncertainty);
int bgnn_tile_valid_counts(bgnn_ctx *ctx, int32_t height, int32_t width, const uint8_t *valid_mask, int32_t n_tiles,
                           const int32_t *origins, int32_t tile_h, int32_t tile_w, int64_t *counts);

/* ---- VR BAG refinement records (next row (f)3: data/vr_bag.py, scripts/inference_native.py main loop) ----
 * `records` is BAG_root/varres_refinements[0, :] as float32 pairs {depth, depth_uncrt} (DEVICE, [n_cells][2]):
 * every refinement grid row-major, grids in varres_metadata.index order (data/vr_bag.py:262-276), i.e. already
 * the concatenated tile layout of bgnn_tiles.
 *
 * bgnn_vr_unpack: depth / uncertainty (may be NULL) planes and the valid mask RefinementGrid.valid_mask defines
 * (data/vr_bag.py:88-92: depth != nodata and finite).  With a grid table (cell_offsets DEVICE int64
 * [n_grids + 1]) it also writes valid_count[g] (DEVICE int64) and keep[g] (DEVICE u8) =
 * valid_count / cells >= min_valid_ratio in float64 (iterate_refinements :293-295) and clears the mask of
 * dropped grids, so that they come out of bgnn_infer_tiles as zeros and are left untouched by bgnn_vr_apply.
 *
 * bgnn_vr_apply: apply_results (scripts/inference_native.py:480-503) on the records, in place: where
 * classification == 2 (noise), mask and confidence >= auto_correct_threshold: depth -= correction,
 * depth_uncrt *= (2 - confidence), float32.  counts (DEVICE u64[3], caller zeroes) += {noise & valid cells,
 * corrected cells, cells whose depth changed (VRBagWriter._corrections_applied, data/vr_bag.py:583-584)};
 * confidence_sum (DEVICE f64[1]) += sum of confidence over valid cells (the log's mean confidence; summed in
 * float64, not in numpy's float32 pairwise order).  Both asynchronous. */
int bgnn_vr_unpack(bgnn_ctx *ctx, const float *records, int64_t n_cells, float nodata, int32_t n_grids,
                   const int64_t *cell_offsets, double min_valid_ratio, float *depth, float *uncertainty,
                   uint8_t *mask, int64_t *valid_count, uint8_t *keep);
int bgnn_vr_apply(bgnn_ctx *ctx, float *records, int64_t n_cells, const uint8_t *mask, const float *classification,
                  const float *confidence, const float *correction, float auto_correct_threshold, uint64_t *counts,
                  double *confidence_sum);

#ifdef __cplusplus
}
#endif
#endif /* BGNN_H */
