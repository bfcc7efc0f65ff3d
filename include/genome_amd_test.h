/*
 * genome_amd_test.h — entry points that exist ONLY in the test build of the library (genome_amd/libgenome_amd_test.so =
 * libgenome_amd.so's objects + csrc/gk_testhooks.o).  The product library exports none of them, reads no environment variable
 * and has no switch a host could flip by accident.  tests/conftest.py points the Python binding at the test build
 * (GK_LIB_PATH); the kernels, the C-ABI of include/genome_amd.h and everything behind it are the same objects in both.
 */
#ifndef GENOME_AMD_TEST_H
#define GENOME_AMD_TEST_H

#include "genome_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test / A-B switches.  Their defaults are read from the environment ONCE, in
 * gk_ctx_create (GK_TEST_NO_RESERVE, GK_HOST_RAGGED, GK_PART_EXACT, GK_GRAPH_UNITIGS=walk|pj); no entry point
 * consults the environment afterwards.  Names: "test_no_reserve", "host_ragged", "part_exact" (0/1),
 * "graph_unitigs" (0 auto, 1 walk, 2 pointer jumping), "graph_walk_queue" (0: one edge per lane), "graph_load_pct" (load factor
 * of the compacted table, percent), "p4_direct" / "fine_exact" / "p2_wide" / "p2_sorted" (-1 auto, 0, 1), "p4_wide" (-1 auto, 0: 4096-key sorts, 1: 8192, 2: 12288),
 * "p45_stripes" (P5 of one stripe of L1 buckets beside P4 of the next), "p24_pieces" (P4 of one piece of a batch beside the L1
 * scatter of the next), "p4_grid" (P4 workgroups per CU), "filter_classic" (1: tombstones + rehash instead of the streaming
 * rebuild), "dist_exchange_ahead" (0: gk_dist_count_routed does not post the next batch's exchange ahead; every rank alike):
 * A/B switches of the kernels in gk_partition.hip / gk_graph.hip / gk_dist.hip; what each measured is in DESIGN.md and
 * profiles/r02.  "min_lnb1" (also GK_MIN_LNB1; 9 / 10: tables of enough segments get 512 / 1024 L1 buckets, the fan-out of
 * tables beyond 34 GB) and "test_max_nb2" (the pipeline refuses tables of more fine buckets per L1 bucket) stage the
 * large-table paths on small tables.
 * Round 3: "graph_mbt" / "graph_mbt_keys" (gk_graph_build on a minimizer-bucketed copy of the table; keys per bucket),
 * "pairs_host" (1: the paired-end walks on host threads), "test_pairs_small_sets" (tiny LDS sets: walks overflow to the host
 * walker), "host_prefetch" (0: a host-fed count does not upload the next chunk beside the current one), "test_max_stage" (bytes
 * per staging area of a host-fed count), "cc_find" (the components' link pass: 3 = look before the CAS, the default; 0 / 1 / 2 =
 * path halving only / no path writes / start node only), "target_load_pct" (load factor new tables are sized for, percent;
 * 0 = 65), and the failure injections of the exchange: "test_dist_small_send" (a send region far too small: re-routed in
 * place), "test_dist_fail_exchange" (this rank's next exchange fails locally with the given code), "test_dist_fail_classify"
 * (this rank cannot stage the queries of its next classified gather). */
int gk_ctx_set_option(gk_ctx *ctx, const char *name, int64_t value);

/* TEST transport: the ranks are threads of ONE process on ONE device; sends, receives and reductions go through a hub in the
 * library (device-to-device copies matched pairwise in posting order) instead of RCCL, which refuses two ranks on one GPU.  Every
 * rank passes the same 128 id bytes (any) and its own context; calls block until the peers have posted the matching operation
 * (an inconsistent order of operations across ranks deadlocks at once; a send and its receive that differ in size are GK_E_COMM).
 * Everything else about the handle is the product code path. */
int gk_dist_create_loopback(gk_ctx *ctx, int rank, int world, const void *id128, gk_dist **out);

#ifdef __cplusplus
}
#endif
#endif
