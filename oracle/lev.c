/*
 * oracle/lev.c -- CPU restatement of the Levenshtein metric on the ANNchor hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under annchor_amd/ may import, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * use it, and only as the checker / timed CPU baseline.
 *
 * What it restates
 * ----------------
 * reference annchor/distances.py:16-20 `levenshtein(x, y)` -> `Levenshtein.distance`
 * (third-party, NOT under /root/reference: python-Levenshtein==0.27.1 ->
 * RapidFuzz==3.13.0, requirements.txt:4,15,16).  Its published semantics are the
 * textbook unit-cost edit distance (insert = delete = substitute = 1), which is
 * fully specified, so two independent restatements are given and cross-checked:
 *
 *   lev_dp()      Wagner-Fischer two-row DP: the definition itself.
 *   lev_myers()   Myers 1999 / Hyyro 2003 multi-word bit-parallel algorithm, the
 *                 published algorithm RapidFuzz uses for long strings.  This is
 *                 the form timed as the CPU baseline ("port") because it is what
 *                 the reference's dependency actually executes.
 *
 * and the batch evaluator restating reference annchor/utils.py:110-177
 * `get_exact(f, X, IJ)`: out[t] = f(X[IJ[t,0]], X[IJ[t,1]]) as float64.
 *
 * Pinned against: reference tests/test_distances.py:9-12 (4 known answers) and
 * tests/test_datasets.py:234-235 ((10,165) -> 299 on the strings data set); see
 * tests/test_oracle_lev.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int lev_dp(const uint8_t *a, int la, const uint8_t *b, int lb)
{
    if (la == 0) return lb;
    if (lb == 0) return la;
    int *row = (int *)malloc(sizeof(int) * (size_t)(lb + 1));
    for (int j = 0; j <= lb; ++j) row[j] = j;
    for (int i = 1; i <= la; ++i) {
        int diag = row[0];
        row[0] = i;
        for (int j = 1; j <= lb; ++j) {
            int up = row[j];
            int sub = diag + (a[i - 1] != b[j - 1]);
            int best = up + 1 < row[j - 1] + 1 ? up + 1 : row[j - 1] + 1;
            row[j] = sub < best ? sub : best;
            diag = up;
        }
    }
    int r = row[lb];
    free(row);
    return r;
}

/* Multi-word Myers/Hyyro.  Pattern = a (length m, split into 64-bit words),
 * text = b.  Tracks the score of the last pattern row while sweeping text. */
int lev_myers(const uint8_t *a, int m, const uint8_t *b, int n)
{
    if (m == 0) return n;
    if (n == 0) return m;
    int words = (m + 63) / 64;
    uint64_t *pm = (uint64_t *)calloc((size_t)256 * words, sizeof(uint64_t));
    uint64_t *vp = (uint64_t *)malloc(sizeof(uint64_t) * words);
    uint64_t *vn = (uint64_t *)malloc(sizeof(uint64_t) * words);
    for (int i = 0; i < m; ++i) pm[(size_t)a[i] * words + i / 64] |= 1ull << (i % 64);
    for (int w = 0; w < words; ++w) { vp[w] = ~0ull; vn[w] = 0; }
    const uint64_t last = 1ull << ((m - 1) % 64);
    int score = m;
    for (int j = 0; j < n; ++j) {
        const uint64_t *pmj = pm + (size_t)b[j] * words;
        uint64_t hp_carry = 1, hn_carry = 0;   /* D[0][j] - D[0][j-1] = +1 */
        for (int w = 0; w < words; ++w) {
            uint64_t x = pmj[w] | hn_carry;
            uint64_t d0 = (((x & vp[w]) + vp[w]) ^ vp[w]) | x | vn[w];
            uint64_t hp = vn[w] | ~(d0 | vp[w]);
            uint64_t hn = d0 & vp[w];
            uint64_t hp_out, hn_out;
            if (w == words - 1) {
                score += (hp & last) != 0;
                score -= (hn & last) != 0;
                hp_out = hn_out = 0;
            } else {
                hp_out = hp >> 63;
                hn_out = hn >> 63;
            }
            hp = (hp << 1) | hp_carry;
            hn = (hn << 1) | hn_carry;
            vp[w] = hn | ~(d0 | hp);
            vn[w] = hp & d0;
            hp_carry = hp_out;
            hn_carry = hn_out;
        }
    }
    free(pm); free(vp); free(vn);
    return score;
}

/* Batch evaluator: strings packed back to back in `chars`, string s occupies
 * chars[offs[s] .. offs[s]+lens[s]).  algo 0 = DP, 1 = Myers.  Returns the
 * number of threads used. */
int lev_pairs(const uint8_t *chars, const int64_t *offs, const int32_t *lens,
              const int64_t *ij, int64_t n, double *out, int algo, int nthreads)
{
    int used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 64)
#endif
    for (int64_t t = 0; t < n; ++t) {
        int64_t i = ij[2 * t], j = ij[2 * t + 1];
        const uint8_t *a = chars + offs[i], *b = chars + offs[j];
        int la = lens[i], lb = lens[j];
        /* shorter string as the bit-vector pattern */
        int d;
        if (algo == 0) d = lev_dp(a, la, b, lb);
        else d = la <= lb ? lev_myers(a, la, b, lb) : lev_myers(b, lb, a, la);
        out[t] = (double)d;
    }
    return used;
}

/* All-pairs brute force, upper triangle, into a dense nx*nx float32 matrix. */
int lev_all_pairs(const uint8_t *chars, const int64_t *offs, const int32_t *lens,
                  int64_t nx, float *dense, int nthreads)
{
    int used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int64_t i = 0; i < nx; ++i) {
        dense[i * nx + i] = 0.0f;
        for (int64_t j = i + 1; j < nx; ++j) {
            const uint8_t *a = chars + offs[i], *b = chars + offs[j];
            int la = lens[i], lb = lens[j];
            int d = la <= lb ? lev_myers(a, la, b, lb) : lev_myers(b, lb, a, la);
            dense[i * nx + j] = dense[j * nx + i] = (float)d;
        }
    }
    return used;
}
