/* Scalar C restatement of oracle/gibbs.py ("DVG block-Gibbs v1").
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used by tests/ as the
 * bit-exact checker at sizes where the numpy version is too slow, and by
 * bench.py's cpu_baseline leg.  The reference has no sampler to restate
 * (its draw is a QPU call, /root/reference/src/utils/common.py:123-138).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp -shared)
 * Every float op below is a single IEEE-754 binary32 operation; -ffp-contract=off
 * forbids fusing them.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define M0 0xD2511F53u
#define M1 0xCD9E8D57u
#define W0 0x9E3779B9u
#define W1 0xBB67AE85u

static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * c[0];
    uint64_t p1 = (uint64_t)M1 * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += W0; k1 += W1;
  }
}

void dvgo_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
  philox4x32_10(c, key[0], key[1]);
  memcpy(out, c, sizeof(c));
}

static inline float spec_exp(float z) {
  const float LOG2E = 1.4426950408889634f;
  const float LN2_HI = 0.693359375f;
  const float LN2_LO = -2.12194440e-4f;
  float k = rintf(z * LOG2E);
  float r = z - k * LN2_HI;
  r = r - k * LN2_LO;
  float p = 1.0f / 720.0f;
  p = p * r; p = p + 1.0f / 120.0f;
  p = p * r; p = p + 1.0f / 24.0f;
  p = p * r; p = p + 1.0f / 6.0f;
  p = p * r; p = p + 0.5f;
  p = p * r; p = p + 1.0f;
  p = p * r; p = p + 1.0f;
  uint32_t bits = (uint32_t)((int32_t)k + 127) << 23;
  float two_k;
  memcpy(&two_k, &bits, 4);
  return p * two_k;
}

float dvgo_spec_exp(float z) { return spec_exp(z); }

/* state: (C, n) int8 row-major, updated in place. */
int dvgo_gibbs(int C, int n, int8_t *state, const uint32_t *chain_ids, const float *hs,
               const float *Js, float beta, const int32_t *order, const int32_t *class_ptr,
               int ncol, const int32_t *adj_ptr, const int32_t *adj_idx, const int32_t *adj_eid,
               uint64_t seed, uint32_t sweep0, int nsweeps) {
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const float two_beta = 2.0f * beta;
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; ++c) {
    int8_t *s = state + (size_t)c * n;
    int8_t newv[4096];
    for (uint32_t t = sweep0; t < sweep0 + (uint32_t)nsweeps; ++t) {
      for (int k = 0; k < ncol; ++k) {
        int lo = class_ptr[k], hi = class_ptr[k + 1];
        for (int a0 = lo; a0 < hi; a0 += 4096) {
          int a1 = a0 + 4096 < hi ? a0 + 4096 : hi;
          for (int a = a0; a < a1; ++a) {
            int i = order[a];
            float f = hs[i];
            for (int q = adj_ptr[i]; q < adj_ptr[i + 1]; ++q) {
              float w = Js[adj_eid[q]];
              f = f + (s[adj_idx[q]] > 0 ? w : -w);
            }
            float z = two_beta * f;
            z = z < -87.0f ? -87.0f : (z > 87.0f ? 87.0f : z);
            float tt = spec_exp(z);
            uint32_t ctr[4] = {(uint32_t)i, chain_ids[c], t >> 2, 0u /* STREAM_GIBBS */};
            philox4x32_10(ctr, k0, k1);
            float u = (float)(ctr[t & 3] >> 8) * (1.0f / 16777216.0f);
            float one_t = 1.0f + tt;
            float b = u * one_t;
            newv[a - a0] = (b < 1.0f) ? 1 : -1;
          }
          /* independent set: write back after the whole class chunk is computed */
          for (int a = a0; a < a1; ++a) s[order[a]] = newv[a - a0];
        }
      }
    }
  }
  return 0;
}

void dvgo_init_state(int C, int n, int8_t *state, const uint32_t *chain_ids, uint64_t seed, uint32_t sweep0) {
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int c = 0; c < C; ++c)
    for (int i = 0; i < n; ++i) {
      uint32_t ctr[4] = {(uint32_t)i, chain_ids[c], sweep0, 1u /* STREAM_INIT */};
      philox4x32_10(ctr, k0, k1);
      state[(size_t)c * n + i] = (ctr[0] >> 31) ? 1 : -1;
    }
}
