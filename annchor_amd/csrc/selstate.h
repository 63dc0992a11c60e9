// selstate.h -- small device-side state blocks shared between the selection (scan.hip) and its two chained consumers: the
// sampler's statistics (features.hip) and the candidate cut (select.hip).  The selection's finishing workgroup writes the
// consumer's state itself (Sel2Epilogue): the one-thread kernels that used to do it (k_sampler_edges, k_cut_set_t) were launches
// of their own between two kernels that wait for each other anyway.
#pragma once
#include "common.h"

struct BinEdges {
    double e[MAXBINS + 1];
    int nb;
};

// ---- the statistics of a sampling step in one round trip (Sampler.get_partition + the bin populations,
// annchor/samplers.py:75-105, utils.py:536-549): quantiles -> bin edges -> bin counts chained on the device
struct SamplerStats {
    BinEdges be;                         // -inf, linspace(q1, q3, nb - 1), +inf
    double q[2];
    unsigned long long counts[MAXBINS];
    int unfinished;                      // the selection could not finish in its tables (the caller takes the general route)
};

struct CutState {
    double t1, t5;          // cut values (prob of the K1-th / K5-th largest)
    int64_t K1, K5;         // requested counts
    int64_t e1, e5;         // how many entries equal to the cut are taken
    int64_t ncand, nnext;
    int all1, all5;         // take every not-computed pair
    // Second key inside the group of pairs whose probability EQUALS the cut (the ECDF takes a few
    // thousand distinct values for ~10^6 pairs, so that group is hundreds to thousands of pairs; the
    // reference's argpartition picks among them arbitrarily).  By position the rest of the budget
    // would all go to the first rows of the pair list; by predicted distance the population the next
    // model is fitted on gets biased (query recall 0.96-0.98 instead of 1.0 on the reference's digits
    // test).  So: a fixed pseudo-random order, ann_tie_scramble(position) ascending, then position.
    // rk = scrambled-position cut inside the group (~0: the whole group is taken); "above the cut" =
    // prob > t || (prob == t && scramble(p) < rk), "on the cut" = prob == t && scramble(p) == rk
    // (taken in position order, e of them).
    unsigned long long rk1, rk5;
    int64_t tie_n1, tie_n5;       // sizes of the two groups
    int64_t tie_gt1, tie_gt5;     // pairs with prob > t
    long long tie_bin1, tie_bin5; // histogram bin of the wanted key (-1: rk is final already)
    long long tie_rem1, tie_rem5; // wanted rank inside the bin (1-based)
    long long tie_len1, tie_len5; // members of the bin
    int64_t tie_got1, tie_got5;   // list cursors
    int tie_overflow;             // a bin's list did not fit TIE_CAP: the host resolves it with the general selection
    int sel_unfinished;           // the cut values came straight from the selection's tables and it could not finish (k_cut_set_t)
};

// What the finishing workgroup of a selection does with its answers besides storing them (kind 0: nothing).
struct Sel2Epilogue {
    int kind;             // 1: sampler statistics -- q, edges, zeroed counts into *st;  2: cut values -- cs_init with t1 / t5 into *cs
    SamplerStats *st;
    int nparts;
    CutState *cs;
    CutState cs_init;
    int need1, need5, force_redo;
};

// np.linspace(q1, q3, num): step = (q3 - q1) / (num - 1), y_i = i * step + q1 (a product and a sum, each rounded; the
// build does not contract them), the last entry q3 itself; a zero step goes through i / div * delta, which gives q1 too.
__device__ __forceinline__ void sel2_epilogue(const Sel2Epilogue &ep, const uint64_t *prefix /*the answers' keys*/, int unfinished)
{
    if (ep.kind == 1) {
        SamplerStats *st = ep.st;
        const int nparts = ep.nparts;
        const double q1 = ann_key_asc_inv(prefix[0]), q3 = ann_key_asc_inv(prefix[1]);
        st->q[0] = q1; st->q[1] = q3;
        const int num = nparts - 1, div = num > 1 ? num - 1 : 1;
        const double delta = q3 - q1, step = delta / (double)div;
        st->be.nb = nparts;
        st->be.e[0] = -INFINITY;
        for (int i = 0; i < num; ++i) {
            double y;
            if (step != 0.0) { const double p = (double)i * step; y = p + q1; }
            else { const double p = ((double)i / (double)div) * delta; y = p + q1; }
            if (num > 1 && i == num - 1) y = q3;
            st->be.e[1 + i] = y;
        }
        st->be.e[nparts] = INFINITY;
        for (int b = 0; b < MAXBINS; ++b) st->counts[b] = 0;
        st->unfinished = unfinished;
    } else if (ep.kind == 2) {
        CutState init = ep.cs_init;
        int q = 0;
        if (ep.need1) init.t1 = ann_key_asc_inv(prefix[q++]);
        if (ep.need5) init.t5 = ann_key_asc_inv(prefix[q++]);
        init.sel_unfinished = unfinished | ep.force_redo;   // (ANNCHOR_CUT_FORCE_REDO: tests walk the second attempt)
        *ep.cs = init;
    }
}
