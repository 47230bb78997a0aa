// Generic (non-stencil) graphs: a PyG `Data` built elsewhere, given as x / edge_index / edge_attr.
// Builds a CSR by target with each row in ascending edge-id order (the order torch_geometric's
// scatter sums in), self loops removed as GATConv does (remove_self_loops before add_self_loops).
#include "bgnn_internal.h"

namespace bgnn {

int launch_generic_build(bgnn_ctx *ctx, bgnn_graph *g, int64_t n_nodes, int32_t n_feat, const float *x,
                         int64_t n_edges, const int64_t *edge_index, int32_t edge_dim, const float *edge_attr) {
  (void)ctx; (void)g; (void)n_nodes; (void)n_feat; (void)x; (void)n_edges; (void)edge_index; (void)edge_dim; (void)edge_attr;
  set_error("bgnn_graph_from_edges: generic graphs are not built yet");
  return BGNN_ERR_UNSUPPORTED;
}

}  // namespace bgnn
