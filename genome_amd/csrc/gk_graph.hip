// placeholder, replaced below
#include "gk_internal.h"
using namespace gk;
extern "C" {
int gk_graph_build(gk_map *m, gk_graph **out) { (void)out; return fail(m ? m->ctx : nullptr, GK_E_STATE, "graph: not built yet"); }
void gk_graph_destroy(gk_graph *) {}
int gk_graph_counts(gk_graph *, uint64_t *, uint64_t *, uint64_t *) { return GK_E_STATE; }
int gk_graph_simplify(gk_graph *) { return GK_E_STATE; }
int gk_graph_remove_bubbles(gk_graph *) { return GK_E_STATE; }
int gk_graph_remove_edges(gk_graph *, const uint64_t *, const uint64_t *, const uint8_t *, uint64_t, uint64_t *) { return GK_E_STATE; }
int gk_graph_retain_largest(gk_graph *, uint64_t *, uint64_t *) { return GK_E_STATE; }
int gk_graph_export_nodes(gk_graph *, uint64_t *, uint64_t *, uint64_t, uint64_t *) { return GK_E_STATE; }
int gk_graph_export_edges(gk_graph *, uint64_t *, uint64_t *, uint64_t *, uint64_t *, int64_t *, int64_t *, uint64_t, uint64_t *, uint8_t *, uint64_t, uint64_t *) { return GK_E_STATE; }
int gk_graph_out_order(gk_graph *, uint64_t, uint64_t, int *, int *) { return GK_E_STATE; }
}
