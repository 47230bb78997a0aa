/* The boundary is plain C: this file includes include/bgnn.h with a C compiler (gcc -std=c99 -pedantic), loads the
 * library with dlopen and resolves every entry point the header declares.  Without a GPU it checks the calls that need
 * none (ABI version, weight count, error text) and that context creation fails cleanly with a message. */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include "bgnn.h"

#define RESOLVE(name)                                             \
  do {                                                            \
    if (!dlsym(lib, #name)) { fprintf(stderr, "missing %s\n", #name); return 2; } \
  } while (0)

int main(int argc, char **argv) {
  void *lib;
  int (*abi)(void);
  size_t (*wcount)(const bgnn_model_desc *);
  int (*ctx_create)(int, void *, bgnn_ctx **);
  const char *(*last_error)(void);
  bgnn_model_desc d;
  bgnn_ctx *ctx = NULL;
  int rc;
  if (argc < 2) return 64;
  lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
  RESOLVE(bgnn_abi_version); RESOLVE(bgnn_last_error); RESOLVE(bgnn_ctx_create); RESOLVE(bgnn_ctx_destroy);
  RESOLVE(bgnn_ctx_synchronize); RESOLVE(bgnn_ctx_stream); RESOLVE(bgnn_ctx_profile); RESOLVE(bgnn_ctx_profile_read);
  RESOLVE(bgnn_model_weight_count); RESOLVE(bgnn_model_create); RESOLVE(bgnn_model_destroy);
  RESOLVE(bgnn_graph_build); RESOLVE(bgnn_graph_from_edges); RESOLVE(bgnn_graph_destroy); RESOLVE(bgnn_graph_counts);
  RESOLVE(bgnn_graph_export); RESOLVE(bgnn_graph_scatter); RESOLVE(bgnn_forward); RESOLVE(bgnn_forward_train); RESOLVE(bgnn_forward_train_dropout); RESOLVE(bgnn_infer_tiles);
  RESOLVE(bgnn_stitch_tiles); RESOLVE(bgnn_cut_tiles); RESOLVE(bgnn_tile_valid_counts); RESOLVE(bgnn_vr_unpack);
  RESOLVE(bgnn_vr_apply); RESOLVE(bgnn_feature_extractor); RESOLVE(bgnn_heads);
  RESOLVE(bgnn_ctx_set_option); RESOLVE(bgnn_ctx_get_option); RESOLVE(bgnn_build_id);
  *(void **)(&abi) = dlsym(lib, "bgnn_abi_version");
  *(void **)(&wcount) = dlsym(lib, "bgnn_model_weight_count");
  *(void **)(&ctx_create) = dlsym(lib, "bgnn_ctx_create");
  *(void **)(&last_error) = dlsym(lib, "bgnn_last_error");
  if (abi() != BGNN_ABI_VERSION) { fprintf(stderr, "ABI %d != header %d\n", abi(), BGNN_ABI_VERSION); return 3; }
  memset(&d, 0, sizeof d);
  d.in_channels = 8; d.hidden = 64; d.num_layers = 4; d.heads = 4; d.num_classes = 3; d.edge_dim = 3;
  d.predict_correction = 1; d.bn_eps = 1e-5f; d.gnn_type = BGNN_GNN_GAT;
  if (wcount(&d) != 182469u + 2u * (256u * 3u + 64u)) { fprintf(stderr, "weight count %lu\n", (unsigned long)wcount(&d)); return 4; }
  rc = ctx_create(0, NULL, &ctx);
  if (rc == BGNN_OK) {
    int (*destroy)(bgnn_ctx *);
    *(void **)(&destroy) = dlsym(lib, "bgnn_ctx_destroy");
    destroy(ctx);
    printf("ok (GPU present)\n");
  } else {
    if (!last_error() || !last_error()[0]) { fprintf(stderr, "no error text after rc=%d\n", rc); return 5; }
    printf("ok (no GPU: %s)\n", last_error());
  }
  return 0;
}
