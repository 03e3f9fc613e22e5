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
  // The fast schedule's LANE-MAJOR image (gibbs.hip::gibbs_fast_kernel), for graphs of at most 20 rows per lane and at
  // most 20 neighbours per spin; null otherwise.  lane_lpc lanes work on a chain (the smallest of 16 / 32 / 64 that
  // covers the largest class in one pass, 64 otherwise).  A class of c spins takes ceil(c / lane_lpc) passes; passes are
  // grouped in STEPS of lane_nr rows (2 when any class takes more than one pass, else 1; a class's last step may hold an
  // empty row), lane_rows = lane_nr * steps rows in all.  Row k of lane l: its spin lane_spin[k lane_lpc + l] (-1: none)
  // and, at entry ((k lane_mb + j) lane_lpc + l) 4 + e, neighbour 4 j + e of that spin: lane_eid = its edge (-1: none),
  // lane_off = the byte offset of its state inside a chain's float16 state row (0 for none).
  int lane_lpc, lane_mb, lane_nr, lane_rows;
  int32_t *lane_spin;            // [lane_rows][lane_lpc]
  int32_t *lane_eid;             // [lane_rows][lane_mb][lane_lpc][4]
  uint16_t *lane_off;            // [lane_rows][lane_mb][lane_lpc][4]
  int32_t h_class_ptr[65];       // host copy (n_colours <= 64)
};
