// Device-side view of the GRBM graph held by a dvg_graph handle.
#pragma once
#include <stdint.h>

struct dvg_graph {
  int n, n_edges, n_colours, max_class, max_degree, n_adj;
  int device;
  // device arrays (owned)
  int32_t *edge_i, *edge_j;      // [n_edges]
  int32_t *order;                // [n] spins in colour-class order
  int32_t *class_ptr;            // [n_colours + 1]
  int32_t *adj_ptr;              // [n + 1]
  int32_t *adj_idx, *adj_eid;    // [2 n_edges]
  // The sampler's LDS image of the neighbour lists (gibbs.hip): every CSR row padded to whole batches of 4 entries.
  int n_batches, max_batches;    // batches over all rows; the longest row's batches
  int32_t *adj_row;              // [n]  (first batch of the row << 8) | its batches
  int32_t *adj_src4;             // [4 n_batches]  CSR position of the entry, -1 for padding
  // The fast schedule's LANE-MAJOR image (gibbs.hip::gibbs_fast_kernel), for graphs of at most 12 (colour class, pass)
  // slots per lane and at most 20 neighbours per spin; null otherwise.  lane_lpc lanes work on a chain (the smallest of
  // 16 / 32 / 64 that covers the largest class in lane_passes <= ... passes); entry ((k lane_mb + j) lane_lpc + l) 4 + e is
  // the CSR position of neighbour 4 j + e of the spin lane l owns in slot k = colour * lane_passes + pass, -1 for none.
  int lane_lpc, lane_mb, lane_passes;
  int32_t *lane_src;             // [n_colours lane_passes][lane_mb][lane_lpc][4]
  int32_t h_class_ptr[65];       // host copy (n_colours <= 64)
};
