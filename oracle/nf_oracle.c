/*
 * CPU oracle in C: float32 restatement of the masked affine coupling chain
 * (variant B: src/real_nvp_toy_densities.py:56-123 + the distrax semantics of
 * SURVEY.md section 8a rows b1-b7) and of the split-halves chain (variant A:
 * src/real_nvp.py:71-171, 251-283).
 *
 * TEST INFRASTRUCTURE ONLY -- used by tests/ (checked against oracle/flows_np.py)
 * and by bench.py's cpu_baseline leg ("port": the same arithmetic on the host's
 * cores, OpenMP over the batch).  Nothing under normalizing_flows_amd/ links or
 * loads this file.  PARITY UNPINNED: see oracle/flows_np.py.
 *
 * Weights use the layer-major packed layout of nf_flow_set_weights
 * (include/nf_hip.h): per layer, per Linear: W [in,out] row-major, then b [out].
 */
#include <math.h>
#include <omp.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define MAXW 1024
#define CLONES __attribute__((target_clones("avx512f", "avx2,fma", "default")))

/* y[out] = x[in] @ W[in,out] + b, optional activation (0 none, 1 relu, 2 leaky 0.01) */
CLONES static void linear(const float* x, const float* W, const float* b, int in, int out, int act, float* y) {
  for (int j = 0; j < out; ++j) y[j] = 0.f;
  for (int k = 0; k < in; ++k) {
    const float xv = x[k];
    const float* w = W + (size_t)k * out;
    for (int j = 0; j < out; ++j) y[j] += xv * w[j];
  }
  for (int j = 0; j < out; ++j) {
    float v = y[j] + b[j];
    if (act == 1) v = v > 0.f ? v : 0.f;
    else if (act == 2) v = v >= 0.f ? v : 0.01f * v;
    y[j] = v;
  }
}

static size_t layer_floats(int first_in, const int* hidden, int nh, int last_out) {
  size_t n = 0;
  int cur = first_in;
  for (int j = 0; j < nh; ++j) { n += (size_t)cur * hidden[j] + hidden[j]; cur = hidden[j]; }
  return n + (size_t)cur * last_out + last_out;
}

/* conditioner MLP on `in` inputs -> `last_out` outputs; returns pointer past the layer's weights */
static const float* mlp(const float* w, const float* x, int first_in, const int* hidden, int nh, int last_out,
                        int act, float* out) {
  float a[MAXW], b[MAXW];
  const float* cur = x;
  int in = first_in;
  float* bufs[2] = {a, b};
  for (int j = 0; j < nh; ++j) {
    linear(cur, w, w + (size_t)in * hidden[j], in, hidden[j], act, bufs[j & 1]);
    w += (size_t)in * hidden[j] + hidden[j];
    cur = bufs[j & 1];
    in = hidden[j];
  }
  linear(cur, w, w + (size_t)in * last_out, in, last_out, 0, out);
  return w + (size_t)in * last_out + last_out;
}

/* Transformed(Independent(Normal(0,1)), Inverse(Chain(MaskedCoupling(ScalarAffine)...))).log_prob */
int nfo_affine_masked_log_prob(const float* weights, int D, int L, const int* hidden, int nh, int first_parity,
                               const float* x, int64_t rows, float* out, int threads) {
  if (D > MAXW || nh < 1) return -1;
  for (int j = 0; j < nh; ++j) if (hidden[j] > MAXW) return -1;
  const size_t per_layer = layer_floats(D, hidden, nh, 2);
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < rows; ++r) {
    float v[MAXW], masked[MAXW], p[2];
    memcpy(v, x + r * D, sizeof(float) * D);
    float ld = 0.f;
    for (int l = L - 1; l >= 0; --l) {          /* Chain.forward: last layer first */
      const int pass = (first_parity + l) & 1;  /* dims with d % 2 == pass go through unchanged */
      int nt = 0;
      for (int d = 0; d < D; ++d) { masked[d] = ((d & 1) == pass) ? v[d] : 0.f; nt += (d & 1) != pass; }
      mlp(weights + per_layer * l, masked, D, hidden, nh, 2, 1, p);
      const float e = expf(p[1]);
      for (int d = 0; d < D; ++d) if ((d & 1) != pass) v[d] = e * v[d] + p[0];
      ld += (float)nt * p[1];
    }
    float ss = 0.f;
    for (int d = 0; d < D; ++d) ss += v[d] * v[d];
    out[r] = -0.5f * ss - (float)D * 0.9189385332046727f + ld;
  }
  return 0;
}

/* log_prob_nvp_chains(params, fns_config, log_prob_N01, y), D even */
int nfo_affine_split_log_prob(const float* weights, int D, int L, const int* hidden, int nh, int first_flip,
                              const float* y, int64_t rows, float* out, int threads) {
  const int d = D / 2;
  if (D > MAXW || (D & 1) || nh < 1) return -1;
  for (int j = 0; j < nh; ++j) if (hidden[j] > MAXW) return -1;
  const size_t per_layer = layer_floats(d, hidden, nh, D);
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < rows; ++r) {
    float v[MAXW], p[MAXW];
    memcpy(v, y + r * D, sizeof(float) * D);
    float ld = 0.f;
    for (int l = L - 1; l >= 0; --l) {
      const int flip = (first_flip + l) & 1;
      float* cond = v + (flip ? d : 0);
      float* tr = v + (flip ? 0 : d);
      mlp(weights + per_layer * l, cond, d, hidden, nh, D, 2, p);
      for (int c = 0; c < d; ++c) { tr[c] = (tr[c] - p[c]) * expf(-p[d + c]); ld -= p[d + c]; }
    }
    float ss = 0.f;
    for (int c = 0; c < D; ++c) ss += v[c] * v[c];
    out[r] = -0.5f * ss - (float)D * 0.9189385332046727f + ld;
  }
  return 0;
}

int nfo_max_threads(void) { return omp_get_max_threads(); }
