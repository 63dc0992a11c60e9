// CPU prototype: transportation simplex (MODI on a spanning-tree basis) on the same reduced problems as emd_sim.cpp, to
// count pivots, cycle lengths and tree depths before writing a wave-level kernel.
//   g++ -O2 -o emd_simplex_sim emd_simplex_sim.cpp && ./emd_simplex_sim data.bin
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
#include <algorithm>
static int NB;
static int BLOCK = 0;   // > 0: block pricing -- most negative arc of the current block of BLOCK row-major arcs, blocks taken round robin
static long PRICED = 0;
static const double *COST;
struct Cnt { long pivots = 0, cyc = 0, solves = 0, maxpiv = 0, depth = 0, degenerate = 0, capped = 0; };
static double solve(const double *hx, const double *hy, Cnt &c, int init_rule)
{
    const int nb = NB;
    double sa = 0, sb = 0;
    for (int k = 0; k < nb; ++k) { sa += hx[k]; sb += hy[k]; }
    std::vector<int> rows, cols; std::vector<long> a, b;
    for (int k = 0; k < nb; ++k) {
        long x = (long)(hx[k] * sb), y = (long)(hy[k] * sa), d = x - y;
        if (d > 0) { rows.push_back(k); a.push_back(d); }
        if (d < 0) { cols.push_back(k); b.push_back(-d); }
    }
    const int n = rows.size(), m = cols.size(), N = n + m;
    if (n == 0) return 0.0;
    auto C = [&](int i, int j) { return COST[rows[i] * nb + cols[j]]; };
    // ---- initial basis: least-cost rule (n + m - 1 arcs, exactly one line eliminated per step)
    std::vector<int> parent(N, -1), depth(N, 0);
    std::vector<long> flow(N, 0);          // flow on the arc between node k and parent[k]
    std::vector<std::vector<std::pair<int,long>>> adj(N);
    {
        std::vector<char> rdone(n, 0), cdone(m, 0);
        std::vector<long> ar = a, br = b;
        int left_r = n, left_c = m;
        while (left_r + left_c > 1) {
            int bi = -1, bj = -1; double best = INFINITY;
            if (init_rule == 0) {
                for (int i = 0; i < n; ++i) if (!rdone[i]) for (int j = 0; j < m; ++j) if (!cdone[j] && C(i, j) < best) { best = C(i, j); bi = i; bj = j; }
            } else {   // row minimum rule: first open row, its cheapest open column
                for (int i = 0; i < n && bi < 0; ++i) if (!rdone[i]) { bi = i; for (int j = 0; j < m; ++j) if (!cdone[j] && C(i, j) < best) { best = C(i, j); bj = j; } }
            }
            if (bi < 0 || bj < 0) break;
            long f = std::min(ar[bi], br[bj]);
            adj[bi].push_back({n + bj, f}); adj[n + bj].push_back({bi, f});
            ar[bi] -= f; br[bj] -= f;
            if (ar[bi] == 0 && (left_r > 1 || left_c == 1) && !(br[bj] == 0 && left_c > 1 && left_r == 1)) { rdone[bi] = 1; --left_r; }
            else { cdone[bj] = 1; --left_c; }
        }
    }
    // root the tree at node 0
    std::vector<double> pot(N, 0.0);
    auto reroot = [&]() {
        std::fill(parent.begin(), parent.end(), -2);
        std::vector<int> st = {0}; parent[0] = -1; depth[0] = 0; pot[0] = 0;
        while (!st.empty()) {
            int k = st.back(); st.pop_back();
            for (auto &e : adj[k]) if (parent[e.first] == -2) {
                int ch = e.first; parent[ch] = k; depth[ch] = depth[k] + 1; flow[ch] = e.second;
                // u_i + v_j = c_ij on tree arcs
                if (ch >= n) pot[ch] = C(k, ch - n) - pot[k]; else pot[ch] = C(ch, k - n) - pot[k];
                st.push_back(ch);
            }
        }
    };
    reroot();
    for (int k = 0; k < N; ++k) if (parent[k] == -2) { fprintf(stderr, "basis is not a spanning tree (n %d m %d)\n", n, m); return NAN; }
    long piv = 0;
    for (;; ++piv) {
        if (piv > 5000) { c.capped++; break; }
        // pricing: most negative reduced cost
        double best = -1e-12; int ei = -1, ej = -1;
        if (BLOCK <= 0) {
            for (int i = 0; i < n; ++i) for (int j = 0; j < m; ++j) { double rc = C(i, j) - pot[i] - pot[n + j]; if (rc < best) { best = rc; ei = i; ej = j; } }
            PRICED += (long)n * m;
        } else {
            static int cur = 0;
            const int nm = n * m, nblk = (nm + BLOCK - 1) / BLOCK;
            if (piv == 0) cur = 0;
            for (int tries = 0; tries < nblk && ei < 0; ++tries) {
                const int b = cur % nblk;
                for (int a = b * BLOCK; a < std::min(nm, (b + 1) * BLOCK); ++a) { int i = a / m, j = a % m; double rc = C(i, j) - pot[i] - pot[n + j]; if (rc < best) { best = rc; ei = i; ej = j; } }
                PRICED += std::min(nm, (b + 1) * BLOCK) - b * BLOCK;
                if (ei < 0) ++cur;
            }
        }
        if (ei < 0) break;
        // cycle: entering arc ei -> n + ej carries +theta; walk both ends up to the common ancestor
        int x = ei, y = n + ej;
        long theta = -1; int leave = -1;   // leaving arc = (leave, parent[leave])
        int len = 1;
        auto consider = [&](int child, bool decreases) { if (decreases && (theta < 0 || flow[child] < theta)) { theta = flow[child]; leave = child; } };
        // flow direction is source -> sink on every arc.  Pushing theta along ei -> ej: the cycle continues from sink ej back to source ei through the tree.
        // On the path from y up: arc (y, parent) with y a sink: flow into y from parent source would DEcrease?  Orient: cycle = ei -> ej (enter), then from ej along tree path to ei.
        // Traversing the tree path from ej to ei: arcs alternate; an arc traversed from sink to source (against flow direction) loses theta, from source to sink gains.
        int xx = x, yy = y;
        while (xx != yy) {
            if (depth[xx] >= depth[yy]) {   // xx side: path ... -> xx (ends at ei): traversed from parent towards xx at the end of the cycle, i.e. direction parent -> xx
                // direction of traversal on this side is from the ancestor DOWN to xx (cycle goes ej ~> lca ~> ei)
                bool child_is_source = xx < n;    // arc between xx (child) and parent: traversed parent -> child; flow direction source -> sink
                // traversed from sink(parent) to source(child) => against the flow => decreases
                consider(xx, child_is_source);
                xx = parent[xx]; ++len;
            } else {                        // yy side: traversed from yy UP to the ancestor (cycle leaves ej upwards)
                bool child_is_sink = yy >= n;     // traversed child -> parent; child sink -> parent source: against the flow => decreases
                consider(yy, child_is_sink);
                yy = parent[yy]; ++len;
            }
        }
        c.cyc += len;
        if (theta == 0) c.degenerate++;
        // apply: update adjacency flows along the cycle, drop the leaving arc, add the entering arc
        auto bump = [&](int u_, int v_, long d) { for (auto &e : adj[u_]) if (e.first == v_) e.second += d; for (auto &e : adj[v_]) if (e.first == u_) e.second += d; };
        xx = x; yy = y;
        while (xx != yy) {
            if (depth[xx] >= depth[yy]) { bump(xx, parent[xx], (xx < n) ? -theta : theta); xx = parent[xx]; }
            else { bump(yy, parent[yy], (yy >= n) ? -theta : theta); yy = parent[yy]; }
        }
        int lp = parent[leave];
        auto drop = [&](int u_, int v_) { auto &A = adj[u_]; for (size_t t = 0; t < A.size(); ++t) if (A[t].first == v_) { A.erase(A.begin() + t); break; } };
        drop(leave, lp); drop(lp, leave);
        adj[ei].push_back({n + ej, theta}); adj[n + ej].push_back({ei, theta});
        reroot();
    }
    int md = 0; for (int k = 0; k < N; ++k) md = std::max(md, depth[k]);
    c.depth += md;
    c.pivots += piv; c.maxpiv = std::max(c.maxpiv, piv); c.solves++;
    double obj = 0;
    for (int k = 1; k < N; ++k) { int p = parent[k]; obj += (double)flow[k] * (k >= n ? C(p, k - n) : C(k, p - n)); }
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
    for (int blk : {0, 512, 256, 128})
    for (int rule = 1; rule < 2; ++rule)
      for (int cls = 0; cls < 2; ++cls) {
        BLOCK = blk; PRICED = 0;
        Cnt c; double maxerr = 0;
        for (int p = 0; p < np; ++p) {
            if ((std::isnan(want[p]) ? 1 : 0) != cls) continue;
            double d = solve(&H[(size_t)ij[2 * p] * nb], &H[(size_t)ij[2 * p + 1] * nb], c, rule);
            if (!std::isnan(want[p])) maxerr = std::max(maxerr, fabs(d - want[p]));
        }
        printf("block %d, arcs priced per solve %.0f | init rule %d %s: pivots per solve %.1f (max %ld), cycle length %.1f, final tree depth %.1f, degenerate pivots %.1f %%, capped %ld | err vs stored %.1e\n",
               BLOCK, (double)PRICED / c.solves, rule, cls ? "far " : "near", (double)c.pivots / c.solves, c.maxpiv, (double)c.cyc / std::max(c.pivots, 1l), (double)c.depth / c.solves,
               100.0 * c.degenerate / std::max(c.pivots, 1l), c.capped, maxerr);
      }
}
