/* OASIS AR(1) deconvolution with a hard minimum spike size (CPU, host side of
 * the post-hoc spike statistics -- SURVEY 8(f) row 2).
 *
 * Restates the published online active-set algorithm of Friedrich, Zhou &
 * Paninski (2017), "Fast online deconvolution of calcium imaging data"
 * (Algorithm 3 + the s_min variant), i.e. what the reference calls through
 * `oasis.oasis_methods.oasisAR1(y, g=0.95, s_min=.55)` in
 * gan/utils/spike_helper.py:23-29.  The OASIS package is an un-pinned git
 * clone in the reference's setup.sh:43 and is not installed here: PARITY
 * UNPINNED; pinned only by the known-answer tests in tests/test_spike_stats.py
 * and by the pure-python restatement kept beside it.
 *
 * Pools (v, w, t, l): value sum, weight sum, start time, length.  A new data
 * point opens a pool; pools are merged backwards while
 *     v[i]/w[i] < g^l[i-1] * v[i-1]/w[i-1] + s_min.
 * c is rebuilt as max(v/w, 0) * g^k inside each pool, s[t] = c[t] - g*c[t-1].
 */
#include <math.h>
#include <stdlib.h>

typedef struct { double v, w; int t, l; } pool_t;

int cg_oasis_ar1(const double* y, int T, double g, double lam, double s_min,
                 double* c, double* s) {
  if (T < 1) return 1;
  pool_t* P = (pool_t*)malloc(sizeof(pool_t) * (size_t)T);
  if (!P) return 2;
  int i = 0;
  P[0].v = y[0] - lam * (1.0 - g);
  P[0].w = 1.0; P[0].t = 0; P[0].l = 1;
  for (int t = 1; t < T; ++t) {
    ++i;
    P[i].v = y[t] - lam * (t == T - 1 ? 1.0 : (1.0 - g));
    P[i].w = 1.0; P[i].t = t; P[i].l = 1;
    while (i > 0 && P[i - 1].v / P[i - 1].w * pow(g, P[i - 1].l) + s_min >
                        P[i].v / P[i].w) {
      --i;
      const double gl = pow(g, P[i].l);
      P[i].v += P[i + 1].v * gl;
      P[i].w += P[i + 1].w * gl * gl;
      P[i].l += P[i + 1].l;
    }
  }
  for (int j = 0; j <= i; ++j) {
    double tmp = P[j].v / P[j].w;
    if (tmp < 0.0) tmp = 0.0;
    for (int k = 0; k < P[j].l; ++k) {
      c[P[j].t + k] = tmp;
      tmp *= g;
    }
  }
  s[0] = 0.0;
  for (int t = 1; t < T; ++t) s[t] = c[t] - g * c[t - 1];
  free(P);
  return 0;
}

/* rows x T signals -> binarised spike trains (spike_helper.py:23-29: s > thr) */
int cg_deconvolve(const double* signals, int rows, int T, double g,
                  double s_min, double threshold, float* spikes) {
  double* c = (double*)malloc(sizeof(double) * (size_t)T * 2);
  if (!c) return 2;
  double* s = c + T;
  for (int r = 0; r < rows; ++r) {
    int rc = cg_oasis_ar1(signals + (size_t)r * T, T, g, 0.0, s_min, c, s);
    if (rc) { free(c); return rc; }
    for (int t = 0; t < T; ++t)
      spikes[(size_t)r * T + t] = s[t] > threshold ? 1.0f : 0.0f;
  }
  free(c);
  return 0;
}
