// CPU model of csrc/emd.hip's solver (int flows, common mass cancelled, one search carries several augmentations) with
// operation counters: searches, pops (wave minima), relaxed source rows, path steps -- to compare start heuristics
// without a GPU.  Input: a binary file written by tools/sim/emd_sim.py (histograms, cost, pairs).
//   g++ -O2 -o emd_sim emd_sim.cpp && ./emd_sim data.bin [variant]
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
#include <algorithm>
#include <cstring>
struct Cnt { long searches = 0, pops = 0, relax = 0, steps = 0, greedy = 0, solves = 0, maxpops = 0, maxwork = 0; double greedy_mass = 0, total_mass = 0; };
static int NB;
static const double *COST;
static double solve(const double *hx, const double *hy, int variant, Cnt &c)
{
    const int nb = NB;
    double sa = 0, sb = 0;
    for (int k = 0; k < nb; ++k) { sa += hx[k]; sb += hy[k]; }
    std::vector<long> xm(nb), ym(nb);
    for (int k = 0; k < nb; ++k) {
        long x = (long)(hx[k] * sb), y = (long)(hy[k] * sa), d = x - y;
        xm[k] = d > 0 ? d : 0; ym[k] = d < 0 ? -d : 0;
    }
    std::vector<int> rows, cols;
    for (int k = 0; k < nb; ++k) { if (xm[k]) rows.push_back(k); if (ym[k]) cols.push_back(k); }
    const int n = rows.size(), m = cols.size();
    std::vector<long> a(n), b(m);
    for (int i = 0; i < n; ++i) a[i] = xm[rows[i]];
    for (int j = 0; j < m; ++j) b[j] = ym[cols[j]];
    std::vector<double> u(n, 0.0), v(m);
    std::vector<int> amin(m, 0);
    auto C = [&](int i, int j) { return COST[rows[i] * nb + cols[j]]; };
    for (int j = 0; j < m; ++j) { double best = INFINITY; for (int i = 0; i < n; ++i) if (C(i, j) < best) { best = C(i, j); amin[j] = i; } v[j] = best; }
    std::vector<long> F((size_t)n * m, 0);
    long tot = 0; for (int i = 0; i < n; ++i) tot += a[i];
    c.total_mass += tot;
    long placed = 0;
    for (int j = 0; j < m; ++j) {
        int i = amin[j]; long f = std::min(a[i], b[j]);
        if (f > 0) { F[i * m + j] = f; a[i] -= f; b[j] -= f; placed += f; }
    }
    if (variant & 1) {
        // row reduction: u_i = min_j (c_ij - v_j); push along the tight arc of every source with supply left
        for (int i = 0; i < n; ++i) {
            double best = INFINITY; int bj = 0;
            for (int j = 0; j < m; ++j) { double r = C(i, j) - v[j]; if (r < best) { best = r; bj = j; } }
            u[i] = best;
            c.greedy++;
        }
        for (int i = 0; i < n; ++i) {
            if (a[i] <= 0) continue;
            for (int j = 0; j < m; ++j) {   // every tight arc of source i
                if (b[j] <= 0 || a[i] <= 0) continue;
                if (C(i, j) - u[i] - v[j] == 0.0) { long f = std::min(a[i], b[j]); F[i * m + j] += f; a[i] -= f; b[j] -= f; placed += f; }
            }
        }
    }
    c.greedy_mass += placed;
    std::vector<double> dist(m), srcdist(n);
    std::vector<int> pred(m), srcfrom(n), prank(m);
    std::vector<char> sinkdone(m), srcdone(n);
    if (variant & 2) {
        // multi-source searches: every source with supply left starts at distance 0; the search ends at the first sink with demand
        // left; one augmentation (bottleneck along the path incl. the start source's supply)
        for (;;) {
            bool any = false;
            for (int i = 0; i < n; ++i) any |= a[i] > 0;
            if (!any) break;
            c.searches++;
            for (int j = 0; j < m; ++j) { dist[j] = INFINITY; pred[j] = -1; sinkdone[j] = 0; }
            std::fill(srcdone.begin(), srcdone.end(), 0);
            for (int i = 0; i < n; ++i) if (a[i] > 0) {
                c.relax++;
                srcdone[i] = 1; srcdist[i] = 0; srcfrom[i] = -1;
                for (int j = 0; j < m; ++j) { double nd = std::max(C(i, j) - u[i] - v[j], 0.0); if (nd < dist[j]) { dist[j] = nd; pred[j] = i; } }
            }
            double mu = 0; int jend = -1;
            for (;;) {
                double best = INFINITY; int js = -1;
                for (int j = 0; j < m; ++j) if (!sinkdone[j] && dist[j] < best) { best = dist[j]; js = j; }
                if (js < 0) break;
                c.pops++;
                sinkdone[js] = 1; mu = best;
                if (b[js] > 0) { jend = js; break; }
                for (int i = 0; i < n; ++i) if (F[i * m + js] > 0 && !srcdone[i]) {
                    c.relax++;
                    srcdone[i] = 1; srcdist[i] = mu; srcfrom[i] = js;
                    for (int j = 0; j < m; ++j) if (!sinkdone[j]) { double nd = std::max(mu + (C(i, j) - u[i] - v[j]), 0.0); if (nd < dist[j]) { dist[j] = nd; pred[j] = i; } }
                }
            }
            if (jend < 0) break;
            for (int i = 0; i < n; ++i) if (srcdone[i]) u[i] += mu - srcdist[i];
            for (int j = 0; j < m; ++j) if (sinkdone[j]) v[j] -= mu - dist[j];
            long delta = b[jend];
            int j = jend, root = -1;
            for (;;) { int i = pred[j]; c.steps++; if (srcfrom[i] < 0) { root = i; break; } delta = std::min(delta, F[i * m + srcfrom[i]]); j = srcfrom[i]; }
            delta = std::min(delta, a[root]);
            for (j = jend;;) { int i = pred[j]; F[i * m + j] += delta; c.steps++; if (srcfrom[i] < 0) break; F[i * m + srcfrom[i]] -= delta; j = srcfrom[i]; }
            a[root] -= delta; b[jend] -= delta;
        }
    } else
    for (int s = 0; s < n; ++s) {
        while (a[s] > 0) {
            c.searches++;
            for (int j = 0; j < m; ++j) { dist[j] = std::max(C(s, j) - u[s] - v[j], 0.0); pred[j] = s; sinkdone[j] = 0; prank[j] = -1; }
            std::fill(srcdone.begin(), srcdone.end(), 0); srcdone[s] = 1; srcdist[s] = 0; srcfrom[s] = -1;
            double mu = 0; int nhit = 0; long need = a[s];
            std::vector<int> order;
            for (;;) {
                double best = INFINITY; int js = -1;
                for (int j = 0; j < m; ++j) if (!sinkdone[j] && dist[j] < best) { best = dist[j]; js = j; }
                if (js < 0) break;
                c.pops++;
                sinkdone[js] = 1; mu = best;
                if (b[js] > 0) { prank[js] = nhit++; order.push_back(js); need -= std::min(need, b[js]); if (need <= 0) break; }
                for (int i = 0; i < n; ++i) if (F[i * m + js] > 0 && !srcdone[i]) {
                    c.relax++;
                    srcdone[i] = 1; srcdist[i] = mu; srcfrom[i] = js;
                    for (int j = 0; j < m; ++j) if (!sinkdone[j]) { double nd = std::max(mu + (C(i, j) - u[i] - v[j]), 0.0); if (nd < dist[j]) { dist[j] = nd; pred[j] = i; } }
                }
            }
            if (nhit == 0) { a[s] = 0; break; }
            for (int i = 0; i < n; ++i) if (srcdone[i]) u[i] += mu - srcdist[i];
            for (int j = 0; j < m; ++j) if (sinkdone[j]) v[j] -= mu - dist[j];
            for (int k = 0; k < nhit; ++k) {
                int jend = order[k];
                long delta = std::min(a[s], b[jend]);
                for (int j = jend; delta > 0;) { int i = pred[j]; if (i == s) break; int jj = srcfrom[i]; delta = std::min(delta, F[i * m + jj]); j = jj; c.steps++; }
                if (delta <= 0) continue;
                for (int j = jend;;) { int i = pred[j]; F[i * m + j] += delta; if (i == s) break; int jj = srcfrom[i]; F[i * m + jj] -= delta; j = jj; c.steps++; }
                a[s] -= delta; b[jend] -= delta;
            }
        }
    }
    double obj = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < m; ++j) obj += (double)F[i * m + j] * C(i, j);
    c.solves++;
    return obj / (sa * sb);
}
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    int hdr[3]; fread(hdr, 4, 3, f);
    const int nx = hdr[0], nb = hdr[1], np = hdr[2];
    NB = nb;
    std::vector<double> H((size_t)nx * nb), cost((size_t)nb * nb), want(np);
    std::vector<int> ij((size_t)np * 2);
    fread(H.data(), 8, H.size(), f); fread(cost.data(), 8, cost.size(), f); fread(ij.data(), 4, ij.size(), f); fread(want.data(), 8, np, f);
    COST = cost.data();
    std::vector<double> base(np);
    for (int variant : {0, 1, 2, 3})
      for (int cls = 0; cls < 2; ++cls) {
        Cnt c; double maxerr = 0, maxdiff = 0;
        for (int p = 0; p < np; ++p) {
            if ((std::isnan(want[p]) ? 1 : 0) != cls) continue;
            long p0 = c.pops, r0 = c.relax;
            double d = solve(&H[(size_t)ij[2 * p] * nb], &H[(size_t)ij[2 * p + 1] * nb], variant, c);
            c.maxpops = std::max(c.maxpops, c.pops - p0);
            c.maxwork = std::max(c.maxwork, (c.pops - p0) * 3 + (c.relax - r0));
            if (!std::isnan(want[p])) maxerr = std::max(maxerr, fabs(d - want[p]));
            if (variant == 0) base[p] = d; else maxdiff = std::max(maxdiff, fabs(d - base[p]));
        }
        printf("variant %d %s: per solve: searches %.1f pops %.1f relaxed rows %.1f path steps %.1f | max pops %ld | greedy %.1f %% | err vs stored %.1e, vs variant 0 %.1e\n",
               variant, cls ? "far " : "near", (double)c.searches / c.solves, (double)c.pops / c.solves, (double)c.relax / c.solves, (double)c.steps / c.solves,
               c.maxpops, 100 * c.greedy_mass / c.total_mass, maxerr, maxdiff);
      }
}
