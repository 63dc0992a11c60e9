/*
 * oracle/emd.c -- CPU restatement of the Wasserstein metric on the ANNchor hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/lev.c header for the rules).
 *
 * What it restates
 * ----------------
 * reference annchor/utils.py:75-86: `wasserstein(x, y) = kantorovich(x, y, cost=M)`
 * with `pynndescent.distances.kantorovich` (third-party, NOT under
 * /root/reference: pynndescent==0.5.13, requirements.txt:12).  Its published
 * behaviour: restrict x and y to their non-zero supports, normalise each to unit
 * mass, and return the optimal value of the transportation LP
 *     min sum_ij F_ij * M[row_i, col_j]   s.t.  F 1 = a,  F^T 1 = b,  F >= 0
 * (solved there by a network simplex).  The optimum VALUE of an LP is unique, so
 * any exact solver restates it up to floating-point rounding; this file uses the
 * successive-shortest-path (primal-dual, Dijkstra on reduced costs) method on the
 * dense bipartite graph, which is also the structure the HIP kernel uses.
 *
 * Pinned against: reference tests/test_datasets.py:107-108
 * (wasserstein(X[10], X[676]) = 0.305587260000565) and the 179 700 exact-EMD
 * distances the reference stores in annchor/data/digits_data.npz
 * ['neighbor_graph'] (produced by the real pynndescent); independently
 * cross-checked against scipy.optimize.linprog(HiGHS) -- tests/test_oracle_emd.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EMD_MAXB 256

double emd_one(const double *x, const double *y, int nb, const double *cost)
{
    int rows[EMD_MAXB], cols[EMD_MAXB];
    int n = 0, m = 0;
    double sa = 0, sb = 0;
    if (nb > EMD_MAXB) return NAN;
    for (int k = 0; k < nb; ++k) {
        if (x[k] != 0) { rows[n++] = k; sa += x[k]; }
        if (y[k] != 0) { cols[m++] = k; sb += y[k]; }
    }
    if (n == 0 || m == 0) return NAN;
    double *C = (double *)malloc(sizeof(double) * (size_t)n * m);
    double *X = (double *)calloc((size_t)n * m, sizeof(double));
    double a[EMD_MAXB], b[EMD_MAXB], u[EMD_MAXB], v[EMD_MAXB];
    double dist[EMD_MAXB], srcdist[EMD_MAXB];
    int pred[EMD_MAXB], srcfrom[EMD_MAXB];
    unsigned char sinkdone[EMD_MAXB], srcdone[EMD_MAXB];
    for (int i = 0; i < n; ++i) { a[i] = x[rows[i]] / sa; u[i] = 0; }
    for (int j = 0; j < m; ++j) b[j] = y[cols[j]] / sb;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) C[i * m + j] = cost[rows[i] * nb + cols[j]];
    for (int j = 0; j < m; ++j) {
        double mn = C[j];
        for (int i = 1; i < n; ++i) if (C[i * m + j] < mn) mn = C[i * m + j];
        v[j] = mn;
    }
    long guard = 64L * (n + m) + 1024;
    int done_all = 0;
    for (int s = 0; s < n && !done_all; ++s) {
        while (a[s] > 0) {
            if (--guard < 0) { free(C); free(X); return NAN; }
            for (int j = 0; j < m; ++j) {
                dist[j] = C[s * m + j] - u[s] - v[j];
                pred[j] = s; sinkdone[j] = 0;
            }
            memset(srcdone, 0, (size_t)n);
            srcdone[s] = 1; srcdist[s] = 0; srcfrom[s] = -1;
            int jend = -1; double mu = 0;
            for (;;) {
                int js = -1; double best = INFINITY;
                for (int j = 0; j < m; ++j)
                    if (!sinkdone[j] && dist[j] < best) { best = dist[j]; js = j; }
                if (js < 0) break;
                sinkdone[js] = 1; mu = best;
                if (b[js] > 0) { jend = js; break; }
                for (int i = 0; i < n; ++i) {
                    if (srcdone[i] || !(X[i * m + js] > 0)) continue;
                    srcdone[i] = 1; srcdist[i] = mu; srcfrom[i] = js;
                    for (int j = 0; j < m; ++j) {
                        if (sinkdone[j]) continue;
                        double nd = mu + (C[i * m + j] - u[i] - v[j]);
                        if (nd < dist[j]) { dist[j] = nd; pred[j] = i; }
                    }
                }
            }
            if (jend < 0) { done_all = 1; break; } /* only rounding dust left */
            for (int i = 0; i < n; ++i) if (srcdone[i]) u[i] += mu - srcdist[i];
            for (int j = 0; j < m; ++j) if (sinkdone[j]) v[j] -= mu - dist[j];
            double delta = a[s] < b[jend] ? a[s] : b[jend];
            for (int j = jend;;) {
                int i = pred[j];
                if (i == s) break;
                int jj = srcfrom[i];
                if (X[i * m + jj] < delta) delta = X[i * m + jj];
                j = jj;
            }
            for (int j = jend;;) {
                int i = pred[j];
                X[i * m + j] += delta;
                if (i == s) break;
                int jj = srcfrom[i];
                X[i * m + jj] -= delta;
                j = jj;
            }
            a[s] -= delta; b[jend] -= delta;
        }
    }
    double total = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) total += X[i * m + j] * C[i * m + j];
    free(C); free(X);
    return total;
}

/* utils.py:110-177 get_exact(f, X, IJ) for f = wasserstein. */
int emd_pairs(const double *X, int64_t nx, int nb, const double *cost,
              const int64_t *ij, int64_t n, double *out, int nthreads)
{
    int used = 1;
    (void)nx;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 16)
#endif
    for (int64_t t = 0; t < n; ++t)
        out[t] = emd_one(X + ij[2 * t] * nb, X + ij[2 * t + 1] * nb, nb, cost);
    return used;
}
