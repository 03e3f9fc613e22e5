// dvg_graph handle: uploads the host-built Gibbs plan (graphs.py::build_plan) once.
#include "common.h"
#include "graph.h"
#include <string.h>
#include <new>

using namespace dvg;

template <typename T>
static int upload(T** dst, const T* src, size_t count) {
  *dst = nullptr;
  if (count == 0) return DVG_OK;
  DVG_CHECK_HIP(hipMalloc((void**)dst, count * sizeof(T)));
  DVG_CHECK_HIP(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
  return DVG_OK;
}

extern "C" int dvg_graph_create(int n, int n_edges, const int32_t* edge_i, const int32_t* edge_j,
                                const int32_t* order, const int32_t* class_ptr, int n_colours,
                                const int32_t* adj_ptr, const int32_t* adj_idx,
                                const int32_t* adj_eid, dvg_graph_t** out) {
  DVG_REQUIRE(out, "graph_create: out is null");
  *out = nullptr;
  DVG_REQUIRE(n > 0 && n_edges >= 0 && n_colours > 0 && n_colours <= 64,
              "graph_create: n=%d n_edges=%d n_colours=%d", n, n_edges, n_colours);
  DVG_REQUIRE(edge_i && edge_j && order && class_ptr && adj_ptr && (n_edges == 0 || (adj_idx && adj_eid)),
              "graph_create: null array");
  DVG_REQUIRE(class_ptr[0] == 0 && class_ptr[n_colours] == n, "graph_create: class_ptr does not cover n");
  DVG_REQUIRE(adj_ptr[0] == 0 && adj_ptr[n] == 2 * n_edges, "graph_create: adj_ptr does not cover 2|E|");
  // validate the colouring: no edge inside a class (the sampler's correctness depends on it)
  {
    int* colour = new (std::nothrow) int[n];
    DVG_REQUIRE(colour, "graph_create: out of host memory");
    for (int i = 0; i < n; ++i) colour[i] = -1;
    bool ok = true;
    for (int k = 0; k < n_colours && ok; ++k)
      for (int p = class_ptr[k]; p < class_ptr[k + 1]; ++p) {
        int v = order[p];
        if (v < 0 || v >= n || colour[v] != -1) { ok = false; break; }
        colour[v] = k;
      }
    for (int e = 0; e < n_edges && ok; ++e) {
      if (edge_i[e] < 0 || edge_i[e] >= n || edge_j[e] < 0 || edge_j[e] >= n) ok = false;
      else if (colour[edge_i[e]] == colour[edge_j[e]]) ok = false;
    }
    delete[] colour;
    DVG_REQUIRE(ok, "graph_create: order/class_ptr is not a proper colouring of the edges");
  }
  dvg_graph* g = new (std::nothrow) dvg_graph;
  DVG_REQUIRE(g, "graph_create: out of host memory");
  memset(g, 0, sizeof(*g));
  g->n = n; g->n_edges = n_edges; g->n_colours = n_colours; g->n_adj = 2 * n_edges;
  g->max_class = 0; g->max_degree = 0;
  for (int k = 0; k < n_colours; ++k) {
    int sz = class_ptr[k + 1] - class_ptr[k];
    if (sz > g->max_class) g->max_class = sz;
    g->h_class_ptr[k] = class_ptr[k];
  }
  g->h_class_ptr[n_colours] = class_ptr[n_colours];
  for (int i = 0; i < n; ++i) {
    int d = adj_ptr[i + 1] - adj_ptr[i];
    if (d > g->max_degree) g->max_degree = d;
  }
  // padded-row image for the sampler (gibbs.hip): row i occupies batches [first, first + ceil(deg / 4))
  int32_t* row = new (std::nothrow) int32_t[n];
  int32_t* src4 = new (std::nothrow) int32_t[(size_t)4 * (n + (size_t)n_edges / 2 + 1)];  // sum of ceil(d/4) <= n + |adj|/4
  if (!row || !src4) { delete[] row; delete[] src4; delete g; DVG_REQUIRE(false, "graph_create: out of host memory"); }
  int nb = 0;
  for (int i = 0; i < n; ++i) {
    const int d = adj_ptr[i + 1] - adj_ptr[i], b = (d + 3) / 4;
    row[i] = (int32_t)(((uint32_t)nb << 8) | (uint32_t)(b > 255 ? 255 : b));
    for (int k = 0; k < 4 * b; ++k) src4[4 * (size_t)nb + k] = k < d ? adj_ptr[i] + k : -1;
    if (b > g->max_batches) g->max_batches = b;
    nb += b;
  }
  g->n_batches = nb;
  hipGetDevice(&g->device);
  int rc = (g->max_batches > 255 || nb >= (1 << 24)) ? DVG_E_UNSUPPORTED : DVG_OK;
  if (rc == DVG_OK) rc = upload(&g->adj_row, row, n);
  if (rc == DVG_OK) rc = upload(&g->adj_src4, src4, (size_t)4 * nb);
  delete[] row; delete[] src4;
  if (rc == DVG_OK) {  // the lane-major image of the sampler's fast schedule (graph.h)
    const int lpc = g->max_class <= 16 ? 16 : g->max_class <= 32 ? 32 : 64;
    const int nr = g->max_class > lpc ? 2 : 1;
    const int mb = g->max_batches <= 4 ? 4 : 5;
    int rows = 0;
    for (int c = 0; c < n_colours; ++c) {
      const int passes = (class_ptr[c + 1] - class_ptr[c] + lpc - 1) / lpc;
      rows += nr * ((passes + nr - 1) / nr);
    }
    if (rows <= (lpc == 64 ? 20 : 12) && g->max_batches <= 5 && n <= 32767) {
      const size_t count = (size_t)rows * mb * lpc * 4;
      int32_t* spin = new (std::nothrow) int32_t[(size_t)rows * lpc];
      int32_t* eid = new (std::nothrow) int32_t[count];
      uint16_t* off = new (std::nothrow) uint16_t[count];
      if (!spin || !eid || !off) {
        delete[] spin; delete[] eid; delete[] off; dvg_graph_destroy(g);
        DVG_REQUIRE(false, "graph_create: out of host memory");
      }
      for (size_t q = 0; q < (size_t)rows * lpc; ++q) spin[q] = -1;
      for (size_t q = 0; q < count; ++q) { eid[q] = -1; off[q] = 0; }
      int k = 0;
      for (int c = 0; c < n_colours; ++c) {
        const int size = class_ptr[c + 1] - class_ptr[c], passes = (size + lpc - 1) / lpc;
        for (int pass = 0; pass < passes; ++pass)
          for (int l = 0; l < lpc && pass * lpc + l < size; ++l) {
            const int i = order[class_ptr[c] + pass * lpc + l], d = adj_ptr[i + 1] - adj_ptr[i];
            spin[(size_t)(k + pass) * lpc + l] = i;
            for (int t = 0; t < d; ++t) {
              const size_t q = (((size_t)(k + pass) * mb + t / 4) * lpc + l) * 4 + t % 4;
              eid[q] = adj_eid[adj_ptr[i] + t];
              off[q] = (uint16_t)(2 * adj_idx[adj_ptr[i] + t]);
            }
          }
        k += nr * ((passes + nr - 1) / nr);
      }
      rc = upload(&g->lane_spin, spin, (size_t)rows * lpc);
      if (rc == DVG_OK) rc = upload(&g->lane_eid, eid, count);
      if (rc == DVG_OK) rc = upload(&g->lane_off, off, count);
      delete[] spin; delete[] eid; delete[] off;
      g->lane_lpc = lpc; g->lane_mb = mb; g->lane_nr = nr; g->lane_rows = rows;
    }
  }
  if (rc != DVG_OK) {
    if (rc == DVG_E_UNSUPPORTED) set_error("graph_create: a spin with more than 1020 neighbours");
    dvg_graph_destroy(g);
    return rc;
  }
  if ((rc = upload(&g->edge_i, edge_i, n_edges)) || (rc = upload(&g->edge_j, edge_j, n_edges)) ||
      (rc = upload(&g->order, order, n)) || (rc = upload(&g->class_ptr, class_ptr, n_colours + 1)) ||
      (rc = upload(&g->adj_ptr, adj_ptr, n + 1)) || (rc = upload(&g->adj_idx, adj_idx, 2 * n_edges)) ||
      (rc = upload(&g->adj_eid, adj_eid, 2 * n_edges))) {
    dvg_graph_destroy(g);
    return rc;
  }
  *out = g;
  return DVG_OK;
}

extern "C" int dvg_graph_destroy(dvg_graph_t* g) {
  if (!g) return DVG_OK;
  hipFree(g->edge_i); hipFree(g->edge_j); hipFree(g->order); hipFree(g->class_ptr);
  hipFree(g->adj_ptr); hipFree(g->adj_idx); hipFree(g->adj_eid); hipFree(g->adj_row); hipFree(g->adj_src4); hipFree(g->lane_spin); hipFree(g->lane_eid); hipFree(g->lane_off);
  delete g;
  return DVG_OK;
}
