/* A complete ANNchor fit through the C-ABI of libannchor_hip.so, from plain C: no Python, no torch, no C++ types in sight.
 * It walks the stages of annchor.Annchor.fit() (reference annchor/annchor.py:532-623) the way annchor_amd/annchor.py does with
 * the order-free DeviceStratifiedSampler and the device-fitted models (nothing but sizes and the graph crosses the boundary),
 * then checks the graph against annchor_brute_force on the same context.
 *
 *   gcc -O2 -I include examples/c_abi_fit.c -L annchor_amd -lannchor_hip -Wl,-rpath,$PWD/annchor_amd -lm -o /tmp/c_abi_fit
 *   /tmp/c_abi_fit [n_strings]
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "annchor_hip.h"

#define CHECK(call)                                                                                              \
    do {                                                                                                         \
        int rc_ = (call);                                                                                        \
        if (rc_ != ANNCHOR_OK) {                                                                                 \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, h ? annchor_last_error(h) : annchor_create_error()); \
            return 1;                                                                                            \
        }                                                                                                        \
    } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void)
{
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(rng_state >> 33);
}
static uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

int main(int argc, char **argv)
{
    annchor_ctx *h = NULL;
    const int64_t nx = argc > 1 ? atoll(argv[1]) : 1200;
    const int32_t n_anchors = 12, k = 10, n_partitions = 7, niters = 2, lookahead = 5, alphabet = 4;
    int64_t n_samples = 2000;
    const double p_work = 0.2;

    /* ---- data: clusters of mutated strings over a 4-letter alphabet, dense symbol codes */
    const int L0 = 96, LMAX = 160;
    uint8_t *sym = malloc((size_t)nx * LMAX);
    int64_t *offs = malloc(sizeof(int64_t) * (size_t)nx);
    int32_t *lens = malloc(sizeof(int32_t) * (size_t)nx);
    for (int64_t s = 0; s < nx; ++s) {
        uint8_t *me = sym + s * LMAX;
        offs[s] = s * LMAX;
        if (s % 100 == 0) {
            lens[s] = L0;
            for (int t = 0; t < L0; ++t) me[t] = (uint8_t)(rnd() % alphabet);
            continue;
        }
        const int64_t par = s - 1 - (int64_t)(rnd() % (uint32_t)(s % 100 < 20 ? s % 100 : 20));   /* an earlier member of the cluster */
        int len = lens[par];
        memcpy(me, sym + par * LMAX, (size_t)len);
        for (int e = 0, ne = 1 + (int)(rnd() % 5); e < ne; ++e) {
            const int p = (int)(rnd() % (uint32_t)len), r = (int)(rnd() % 10);
            if (r < 4) me[p] = (uint8_t)(rnd() % alphabet);
            else if (r < 7 && len > L0 / 2) { memmove(me + p, me + p + 1, (size_t)(len - p - 1)); --len; }
            else if (len < LMAX - 1) { memmove(me + p + 1, me + p, (size_t)(len - p)); me[p] = (uint8_t)(rnd() % alphabet); ++len; }
        }
        lens[s] = len;
    }

    /* ---- the fit */
    CHECK(annchor_create(0, &h));
    CHECK(annchor_set_strings(h, sym, offs, lens, nx, alphabet));
    const double N = (double)nx * (double)(nx - 1) / 2.0;
    int64_t na = 0;
    for (int j = 1; j <= n_anchors; ++j) na += nx - j;                       /* annchor.py:129: anchor evaluations */
    CHECK(annchor_pick_anchors_maxmin(h, n_anchors, 0));                      /* pickers.py:28-59 */
    int64_t n_pairs = 0, min_len = 0;
    CHECK(annchor_build_locality(h, 5, 1, (int32_t)(nx < 300 ? nx : 300), &n_pairs, &min_len));   /* annchor.py:208-256 */
    CHECK(annchor_compute_features(h));                                       /* annchor.py:258-311 */
    int64_t evals = na;
    for (int it = 0; it < niters; ++it) {
        /* sampling step (samplers.py:113-140 with the order-free choice) */
        int64_t n_unc = 0;
        CHECK(annchor_count_uncomputed(h, &n_unc));
        int64_t ks[2] = {(int64_t)((double)n_unc / 100.0), (int64_t)(99.0 * (double)n_unc / 100.0)};
        if (ks[0] * n_partitions < n_samples) { ks[0] = (int64_t)((double)n_unc / 10.0); ks[1] = (int64_t)(9.0 * (double)n_unc / 10.0); }
        if (ks[0] * n_partitions < n_samples) n_samples = ks[0] * n_partitions;
        double q[2], edges[65];
        int64_t counts[64], want[64];
        int32_t fused = 0;
        CHECK(annchor_sampler_stats(h, ks, n_partitions, q, edges, counts, &fused));
        if (!fused) {   /* (very long lists: the edges here, the counts by their own call) */
            edges[0] = -INFINITY; edges[n_partitions] = INFINITY;
            for (int b = 0; b < n_partitions - 1; ++b) edges[1 + b] = q[0] + (double)b * ((q[1] - q[0]) / (double)(n_partitions - 2));
            edges[n_partitions - 1] = q[1];
            CHECK(annchor_bin_counts(h, edges, n_partitions, counts));
        }
        for (int b = 0; b < n_partitions; ++b) want[b] = n_samples / n_partitions + (b < n_samples % n_partitions);
        int64_t m = 0;
        CHECK(annchor_hash_sample_pairs_device(h, edges, n_partitions, counts, want, splitmix64(42u + (uint64_t)it), &m));
        evals += m;
        /* models (regressors.py:39-103, error_predictors.py:26-67), fitted where the samples are */
        CHECK(annchor_fit_regression_device(h, edges, n_partitions, it == 0, 1));
        const int32_t nmin = it == 0 ? 3 * k / 2 : 0;
        CHECK(annchor_select_prepare(h, k, nmin));
        CHECK(annchor_fit_errors_device(h));
        /* selection + refinement (annchor.py:395-473) */
        int64_t n_refine = (int64_t)((p_work * N - (double)na - (double)n_samples) / niters) + 1;
        if (n_refine < 0) n_refine = 0;
        int64_t n_cand = 0, n_next = 0;
        CHECK(annchor_select_candidates(h, k, nmin, NULL, NULL, n_partitions, n_refine, lookahead, &n_cand, &n_next));
        CHECK(annchor_refine_candidates(h));
        evals += n_cand;
        if (it < niters - 1) CHECK(annchor_update_bounds(h));                 /* annchor.py:475-512 */
    }
    int64_t *ng_idx = malloc(sizeof(int64_t) * (size_t)nx * k);
    double *ng_dist = malloc(sizeof(double) * (size_t)nx * k);
    CHECK(annchor_neighbor_graph(h, k, ng_idx, ng_dist));                     /* annchor.py:514-530 */
    double W[64 * 3], c[64];
    int32_t status[64], flags[3];
    int64_t err_ptr[65];
    CHECK(annchor_model_download(h, W, c, status, err_ptr, flags));
    if (flags[0] || flags[1] || flags[2]) { fprintf(stderr, "device model flags %d %d %d\n", flags[0], flags[1], flags[2]); return 2; }

    /* ---- the exact graph on the same context, and the comparison compare_neighbor_graphs makes (annchor.py:1047-1066) */
    int64_t *bf_idx = malloc(sizeof(int64_t) * (size_t)nx * k);
    double *bf_dist = malloc(sizeof(double) * (size_t)nx * k);
    CHECK(annchor_brute_force(h, k, bf_idx, bf_dist));
    int64_t errors = 0;
    for (int64_t i = 0; i < nx; ++i) {
        const double kth = bf_dist[i * k + k - 1];
        for (int t = 0; t < k; ++t) errors += ng_dist[i * k + t] > kth;      /* a listed neighbour farther than the true k-th */
    }
    printf("c_abi_fit: %lld strings, %lld candidate pairs, %lld metric evaluations (%.1f %% of all pairs), %lld errors of %lld, W[0]=(%g %g %g)\n",
           (long long)nx, (long long)n_pairs, (long long)evals, 100.0 * (double)evals / N, (long long)errors, (long long)(nx * k), W[0], W[1], W[2]);
    annchor_destroy(h);
    return errors <= nx * k / 50 ? 0 : 3;
}
