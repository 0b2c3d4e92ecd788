/*
 * fista_ref.c -- plain-C twin of oracle/fista.py + oracle/penalty.py.  TEST INFRASTRUCTURE ONLY
 * (see oracle/__init__.py): used by tests as a second checker and by bench.py's `cpu_baseline`
 * leg as the host-core baseline.  Never linked into the product library.
 *
 * Restates what the reference obtains from cvxpy's `problem.solve` at
 * /root/reference/src/sparselm/model/_base.py:516-518 for the objective of
 * model/_lasso.py:109-121 (loss 1/(2n)||X b - y||^2) with the penalty family
 *   sum_j a_j|b_j| + sum_g b_g||b_g||_2 + 1/2 sum_g d_g||b_g||_2^2
 * (model/_lasso.py:99-107, 267-275, 627-639, 795-811).
 *
 * Build: gcc -O3 -march=x86-64-v3 -fopenmp -shared -fPIC fista_ref.c -o _build/libfista_ref.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* g = X^T (w .* (X z - y)) / n in ONE pass over row-major X (n x p, leading dim ld);
 * returns loss = 1/(2n) sum_i w_i (x_i z - y_i)^2.  w may be NULL (all ones). */
double oracle_gradient(const double* X, int64_t n, int64_t p, int64_t ld, const double* y,
                       const double* w, const double* z, double* g) {
  int nthreads = 1;
#ifdef _OPENMP
  nthreads = omp_get_max_threads();
#endif
  double* part = (double*)calloc((size_t)nthreads * (size_t)p, sizeof(double));
  double* lpart = (double*)calloc((size_t)nthreads, sizeof(double));
#pragma omp parallel
  {
    int t = 0;
#ifdef _OPENMP
    t = omp_get_thread_num();
#endif
    double* gp = part + (size_t)t * (size_t)p;
    double lacc = 0.0;
#pragma omp for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      const double* row = X + i * ld;
      double s = 0.0;
      for (int64_t j = 0; j < p; ++j) s += row[j] * z[j];
      const double e = s - y[i];
      const double r = w ? w[i] * e : e;
      lacc += r * e;
      for (int64_t j = 0; j < p; ++j) gp[j] += r * row[j];
    }
    lpart[t] = lacc;
  }
  const double inv_n = 1.0 / (double)n;
  double loss = 0.0;
  for (int64_t j = 0; j < p; ++j) {
    double s = 0.0;
    for (int t = 0; t < nthreads; ++t) s += part[(size_t)t * (size_t)p + j];
    g[j] = s * inv_n;
  }
  for (int t = 0; t < nthreads; ++t) loss += lpart[t];
  free(part);
  free(lpart);
  return 0.5 * loss * inv_n;
}

/* NUMA-friendly copy of a row-major matrix: pages are first touched by the thread that will stream
 * them in oracle_gradient (same static row schedule), so a multi-socket host reads X from all of its
 * memory controllers instead of the one node a single-threaded allocation landed on.  Free with
 * oracle_free. */
double* oracle_numa_copy(const double* src, int64_t n, int64_t p) {
  double* dst = (double*)malloc(sizeof(double) * (size_t)n * (size_t)p);
  if (!dst) return 0;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) memcpy(dst + i * p, src + i * p, sizeof(double) * (size_t)p);
  return dst;
}

void oracle_free(double* ptr) { free(ptr); }

static double soft(double v, double thr) {
  const double m = fabs(v) - thr;
  return m <= 0.0 ? 0.0 : copysign(m, v);
}

/* out = prox_{step * penalty}(v) */
void oracle_prox(const double* v, int64_t p, double step, const double* a, const double* b,
                 const double* d, const int32_t* gidx, int32_t G, double* out, double* work_G) {
  memset(work_G, 0, sizeof(double) * (size_t)G);
  for (int64_t j = 0; j < p; ++j) {
    out[j] = soft(v[j], step * a[j]);
    work_G[gidx[j]] += out[j] * out[j];
  }
  for (int32_t g = 0; g < G; ++g) {
    const double nrm = sqrt(work_G[g]);
    double sc = nrm > 0.0 ? fmax(0.0, 1.0 - step * b[g] / nrm) : 0.0;
    work_G[g] = sc / (1.0 + step * d[g]);
  }
  for (int64_t j = 0; j < p; ++j) out[j] *= work_G[gidx[j]];
}

/* FISTA with gradient-scheme restart; same iteration and stopping rule as oracle/fista.py.
 * beta: in = warm start, out = solution.  Returns iterations used (negative if not converged). */
int64_t oracle_fista(const double* X, int64_t n, int64_t p, int64_t ld, const double* y,
                     const double* w, const double* a, const double* b, const double* d,
                     const int32_t* gidx, int32_t G, double L, double tol, int64_t max_iter,
                     int restart, double* beta) {
  double* z = (double*)malloc(sizeof(double) * (size_t)p);
  double* g = (double*)malloc(sizeof(double) * (size_t)p);
  double* v = (double*)malloc(sizeof(double) * (size_t)p);
  double* bn = (double*)malloc(sizeof(double) * (size_t)p);
  double* wg = (double*)malloc(sizeof(double) * (size_t)(G > 0 ? G : 1));
  memcpy(z, beta, sizeof(double) * (size_t)p);
  const double step = 1.0 / L;
  double t = 1.0;
  int64_t it = 0;
  int converged = 0;
  for (it = 1; it <= max_iter; ++it) {
    oracle_gradient(X, n, p, ld, y, w, z, g);
    for (int64_t j = 0; j < p; ++j) v[j] = z[j] - step * g[j];
    oracle_prox(v, p, step, a, b, d, gidx, G, bn, wg);
    double r2 = 0.0, b2 = 0.0, dotp = 0.0;
    for (int64_t j = 0; j < p; ++j) {
      const double dz = bn[j] - z[j];
      r2 += dz * dz;
      b2 += bn[j] * bn[j];
      dotp += -dz * (bn[j] - beta[j]);
    }
    if (restart && dotp > 0.0) t = 1.0;
    const double t_new = 0.5 * (1.0 + sqrt(1.0 + 4.0 * t * t));
    const double mom = (t - 1.0) / t_new;
    for (int64_t j = 0; j < p; ++j) {
      z[j] = bn[j] + mom * (bn[j] - beta[j]);
      beta[j] = bn[j];
    }
    t = t_new;
    if (sqrt(r2) <= tol * sqrt(b2)) {
      converged = 1;
      break;
    }
  }
  free(z); free(g); free(v); free(bn); free(wg);
  return converged ? it : -(it > max_iter ? max_iter : it);
}
