// euclid.hip -- Euclidean distance over a pair list (the pair-list form of the path).
//
// Replaces, for f = euclidean (reference annchor/distances.py:8-13,
// `np.linalg.norm(x - y)` in the dtype of X, stored into a float64 array) and f = cosine
// (annchor/utils.py:14,67: scipy.spatial.distance.cosine), the evaluator
// get_exact(f, X, IJ) of annchor/utils.py:110-177.
//
// A pair is a gather of two rows, so the kernel is gather/HBM bound: 16 lanes
// cooperate on one pair with 16-byte loads when the row stride allows it, the
// squared differences are accumulated in float64 and the result is rounded to the
// input precision (what a correctly rounded float32 norm returns) before it is
// widened to float64.  The anchor GEMM form for large N lives in euclid_gemm.hip.
#include "common.h"

#define EU_LPP 16  // lanes per pair

template <typename T> struct EuArgs {
    const T *X;
    int dim;
    const int2 *ij;
    const int32_t *idx;
    const int32_t *anchor;
    int64_t n;
    double *out;
    double *RA;
    uint8_t *ncm;
};

template <typename T, int VEC, bool COS> __global__ __launch_bounds__(256) void k_euclid(EuArgs<T> a)
{
    const int sub = threadIdx.x & (EU_LPP - 1);
    const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / EU_LPP;
    const bool active = t < a.n;
    int i = 0, j = 0;
    int64_t opos = t;
    if (active) {
        if (a.anchor) { i = *a.anchor; j = (int)t; }
        else {
            int64_t q = a.idx ? a.idx[t] : t;
            int2 p = a.ij[q];
            i = p.x; j = p.y;
            if (a.idx) opos = q;
        }
    }
    const T *xi = a.X + (size_t)i * a.dim, *xj = a.X + (size_t)j * a.dim;
    double acc = 0, uu = 0, vv = 0;   // Euclidean: sum of squared differences; cosine: u.v, u.u, v.v
    auto term = [&](double x, double y) {
        if (COS) { acc += x * y; uu += x * x; vv += y * y; }
        else { const double d = x - y; acc += d * d; }
    };
    if (active) {
        if (VEC > 1) {
            struct alignas(sizeof(T) * VEC) V { T v[VEC]; };
            const int nv = a.dim / VEC;
            for (int k = sub; k < nv; k += EU_LPP) {
                V u = reinterpret_cast<const V *>(xi)[k], w = reinterpret_cast<const V *>(xj)[k];
#pragma unroll
                for (int e = 0; e < VEC; ++e) term((double)u.v[e], (double)w.v[e]);
            }
        } else {
            for (int k = sub; k < a.dim; k += EU_LPP) term((double)xi[k], (double)xj[k]);
        }
    }
#pragma unroll
    for (int off = EU_LPP / 2; off > 0; off >>= 1) {
        acc += __shfl_xor(acc, off, EU_LPP);
        if (COS) { uu += __shfl_xor(uu, off, EU_LPP); vv += __shfl_xor(vv, off, EU_LPP); }
    }
    if (active && sub == 0) {
        double d;
        if (COS) {
            // scipy.spatial.distance.cosine: the three dot products in the dtype of X, then
            // 1 - uv / sqrt(uu * vv) in float64, clipped to [0, 2]
            if (sizeof(T) == 4) { acc = (double)(float)acc; uu = (double)(float)uu; vv = (double)(float)vv; }
            d = 1.0 - acc / sqrt(uu * vv);
            d = fmin(fmax(d, 0.0), 2.0);
        } else {
            d = sqrt(acc);
            if (sizeof(T) == 4) d = (double)(float)d;
        }
        if (a.out) a.out[t] = d;
        if (a.RA) { a.RA[opos] = d; a.ncm[opos] = 0; }
    }
}

template <typename T, bool COS> static int launch(annchor_ctx *c, const PairSource &src, double *d_out, double *d_RA, uint8_t *d_ncm)
{
    EuArgs<T> a;
    a.X = c->pts.as<T>();
    a.dim = c->dim;
    a.ij = src.ij; a.idx = src.idx; a.anchor = src.anchor; a.n = src.n;
    a.out = d_out; a.RA = d_RA; a.ncm = d_ncm;
    const int vec = 16 / (int)sizeof(T);
    int blocks = ann_blocks(src.n * EU_LPP, 256);
    ProfScope ps(c, COS ? "cosine_pairs" : "euclidean_pairs", (double)src.n * (2.0 * c->dim * sizeof(T) + 16));
    if (c->dim % vec == 0) k_euclid<T, 16 / sizeof(T), COS><<<blocks, 256, 0, c->stream>>>(a);
    else k_euclid<T, 1, COS><<<blocks, 256, 0, c->stream>>>(a);
    ANN_CHECK_HIP(c, hipGetLastError());
    return ANNCHOR_OK;
}

int ann_euclid_launch(annchor_ctx *c, const PairSource &src, double *d_out, double *d_RA, uint8_t *d_ncm)
{
    if (src.n == 0) return ANNCHOR_OK;
    switch (c->metric) {
    case ANNCHOR_METRIC_EUCLIDEAN_F32: return launch<float, false>(c, src, d_out, d_RA, d_ncm);
    case ANNCHOR_METRIC_COSINE_F32: return launch<float, true>(c, src, d_out, d_RA, d_ncm);
    case ANNCHOR_METRIC_COSINE_F64: return launch<double, true>(c, src, d_out, d_RA, d_ncm);
    default: return launch<double, false>(c, src, d_out, d_RA, d_ncm);
    }
}
