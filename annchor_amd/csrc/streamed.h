// streamed.h -- state and argument blocks shared by streamed.hip (tile kernels, ordering) and
// sharded.hip (the device-resident multi-rank protocol around them).
#pragma once
#include "common.h"

#define ST_T 128
#define ST_SLAB 32
#define ST_THREADS 256
#define ST_KMAX 32
#define ST_KMAX_BIG 64
#define ST_KMAX_HUGE 128   // exact-f32 kernels only (dims <= 256): two workgroups per row tile, each keeping the lists of 64 rows
#define ST_SURV 1024
#define ST_KEEP 512
#define ST_EARLY_WINDOW 64   // tiles per yield window of the tile phase (knn_tile_phase)
#define ST_CL_CAP 4096        // entries of a row tile's short list of column tiles (knnbf.hip)
#define ST_CL_TARGET 2048     // eligible tiles a rebuilt short list aims for (4 ST_KEEP)

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef ST_PROFILE
#define ST_PROF_DECL long long pf_t = clock64(); long long pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define ST_PROF(i) { const long long pf_n = clock64(); pf[i] += pf_n - pf_t; pf_t = pf_n; }
#else
#define ST_PROF_DECL
#define ST_PROF(i)
#endif

// LDS hand-over between lanes of ONE wavefront (a wave's LDS instructions execute in order)
__device__ __forceinline__ void wave_fence_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

struct StreamState {
    // local shard (this context's rows), all device memory owned by the context
    DevBuf X;        // float [n_local][dim]      (copy of the caller's rows)
    DevBuf keys, keys2, vals, vals2, cubtmp;
    DevBuf Xs;       // float [n_pad][dimp]       rows in tile order, zero padded
    DevBuf Xb;       // fp16 [n_pad][2][dimp]     the same rows, centred and scaled, split into hi + lo halves (knnbf.hip, knnbk.hip)
    DevBuf rsb;      // float [n_pad]             squared norms of the centred, scaled rows (+inf on padding rows)
    DevBuf cvec;     // float [dimp + 2]          the centre: mean of the anchors' coordinates (zeros without anchors); [dimp] the power-of-two scale
    DevBuf rs;       // float [n_pad]             squared norms (+inf on padding rows)
    DevBuf perm;     // int64 [n_pad]             global id of each ordered row (-1 padding)
    DevBuf lo, hi, mid;  // float [na][nt]        per-tile anchor-distance intervals and means
    DevBuf avec;     // float [dimp]              current anchor vector
    DevBuf runmin, red_val, red_idx;
    DevBuf D;        // float [na][n_local]       distances to anchors (f32)
    DevBuf Dt;       // float [n_local][na padded to 4]  the same, point-major (ordering gathers)
    DevBuf out_d2, out_col;
    DevBuf emit_idx, emit_dist;   // int64 / double [n_local][k]: graph rows in shard order
    DevBuf scr_key, scr_lb;   // float [tile_count][nt_all]: per-row-tile rank keys / bounds of all column tiles
    DevBuf scr_cl;            // uint32 [tile_count][3][ST_CL_CAP]: per-row-tile short lists of the selection rounds
    DevBuf evals;
    DevBuf eval_bits;         // uint32 [tile_count][ceil(nt_all / 32)]: column tiles the tile phase evaluated, per row tile
    DevBuf out_d2b, out_colb; // second list buffers: a join pass reads the old lists of ALL rows and writes new ones
    DevBuf ucand, ucount;     // uint32 [tile_count][JN_CAP] / int32 [tile_count]: join candidates per row tile
    DevBuf rev_cnt, rev_ptr, rev_edges;   // scratch of the reverse neighbour lists (join passes): counts, CSR pointers, edge records of the own columns
    int64_t n_local = 0, n_pad = 0, base = 0;
    int64_t last_tile_evals = 0, last_join_chunks = 0, last_fetched_tiles = 0;
    int last_kernel = 0;   // tile phase of the last build: 0 k_st_knn (exact f32), 1 k_st_knnbf (split fp16)
    int64_t last_guard_rows = 0;   // rows the split-fp16 kernel flagged (error band of the split products reaches the list boundary)
    bool last_repaired = false;    // the flagged rows were done again exactly (repair.hip)
    bool last_two_stage = false;   // the tile phase ran k_st_knnh (knnh.hip) behind its warm-up
    DevBuf guard_tiles;            // uint32 [tile_count][4]: bitmask of the flagged rows of every row tile
    DevBuf guard_list;             // uint32 [1 + flagged]: their number, then (row tile << 7 | row) of each (repair.hip)
    int dim = 0, dimp = 0, na = 0, nt = 0;
    struct KnnArgs *run = nullptr;   // arguments of the graph build in progress (begin / join / end)
    const void *run_perm = nullptr;
    int run_dimp = 0;
    bool run_finished = false;   // annchor_stream_knn_run: the finished graph waits on the device (fin_idx / fin_dist) for _fetch
    int64_t *fin_idx = nullptr;
    float *fin_dist = nullptr;
    // ---- device-resident multi-rank protocol (sharded.hip)
    DevBuf cand_all;   // double [world][2 + dim]: all-gather target of the candidates
    DevBuf cand;       // double [2 + dim]: this rank's arg-max candidate of a max-min round (value, global row, its coordinates)
    DevBuf avecs;      // float [na][dim]: the anchors' coordinates (every rank holds all of them)
    DevBuf A_dev;      // int64 [na]: global row ids of the anchors
    DevBuf rows_send, rows_recv, rows_all;   // raw-row exchange: padded shard, [world][most][dim], compacted [n_total][dim]
    DevBuf lists_all;  // int32 [n_all][K]: every rank's neighbour lists (all-gather target of the join passes)
    DevBuf route_tab, route_cnt, route_slot, route_send, route_recv;   // finished rows on their way to their owners
    DevBuf order_all;  // uint32 [n_pad]: the ranks' slices of the k-d order (all-gather target of annchor_stream_order_begin)
    uint32_t *order_cur = nullptr;     // the order buffer annchor_stream_order_begin finished in (vals or vals2)
    int order_tile_begin = 0, order_tile_count = 0;
    DevBuf rev_all;    // int32 [n_all][JN_RK]: the ranks' reverse-list slices (all-gather target of annchor_stream_join_rev_begin)
    DevBuf rev_slice;  // int32 [tile_count x 128][JN_RK]: this rank's slice (the all-gather's input)
    bool rev_gathered = false;         // rev_all holds the reverse lists of the lists the next join pass is given
    DevBuf D_send, D_recv;   // float [na][most] / [world][na][most]: anchor distances of the own rows on their way to every rank
    bool D_gathered = false;
    int64_t own_base = 0, own_n = 0;   // the shard this context was bound to before it took every rank's rows
    int64_t emit_rows = 0;             // rows of emit_idx / emit_dist (own_n padded to the largest shard)
    int emit_k = 0;
};

struct KnnArgs {
    const float *Xs;      // [n_all][DIM]   all column tiles (every rank's ordered shard, concatenated)
    const uint16_t *Xb;   // [n_all][2][DIM] fp16 hi / lo halves of scale * (the same rows - cvec) (NULL: none -- the f32 kernel runs)
    const float *rsb;     // [n_all] squared norms of the centred rows
    const float *cvec;    // [DIM + 1] what was subtracted, then the scale (the row operand is centred and scaled in the kernel)
    const float *rs;      // [n_all]
    const float *lo, *hi, *mid; // [na][nt_all]
    int nt_all, na;
    // row tiles: the same arrays for the k-NN graph; a separate (query) set for annchor_stream_query
    const float *Rs, *rr, *rlo, *rhi, *rmid;
    int nt_r;             // tile count of the row tables (stride of rlo / rhi / rmid)
    int query;            // 1: rows are queries -- no self tile, no self exclusion
    int tile_begin;       // first row tile of this launch inside the row tile numbering
    int tile_count;
    int K;                // neighbours kept per row, self excluded
    int max_tiles;        // column-tile budget per row tile
    float *out_d2;        // [tile_count*128][K]
    int32_t *out_col;     // [tile_count*128][K]   global ordered column index
    float *scr_key, *scr_lb;   // [tile_count][nt_all] per-row-tile rank keys / valid bounds of every column tile
    unsigned long long *evals;
    unsigned long long *prof;   // ST_PROFILE builds only: per-phase cycle sums (8 counters)
    uint32_t *eval_bits;  // [tile_count][eval_words] evaluated column tiles per row tile (NULL: not recorded)
    int eval_words;
    int eval_halves;      // bitmap rows per row tile: 2 when the lists are split over two workgroups (n_neighbors > 65), else 1
    // join passes (k_st_join_cands / k_st_join)
    const int32_t *lists_all;   // [n_all][K] current neighbour lists of EVERY ordered row (all ranks)
    const uint32_t *ucand;      // [tile_count][ucap] sorted candidate columns per row tile, 0xffffffff padded to 128
    const int32_t *ucount;      // [tile_count]
    int ucap;
    float *out_d2_new;          // [tile_count*128][K] lists after the pass
    int32_t *out_col_new;
    unsigned long long *updates; // list insertions of the pass (its yield: the host stops when it dries up)
    int early_window, early_tau; // tile phase: stop a row tile when early_window consecutive tiles made < early_tau insertions (0: never)
    int dimr;                    // knnbk.hip: the rows' padded dimension (a multiple of 128), set by its launcher
    uint32_t *scr_cl;            // [tile_count][3][ST_CL_CAP] per row tile: the selection rounds' short list (NULL: the rounds sweep the scratch rows)
    int pre_ranked;              // scr_key / scr_lb already hold every (row tile, column tile) pair's rank key and bound (k_st_rank_pairs)
    uint32_t *guard_tiles;       // [tile_count][4] bitmask of the rows the split-fp16 kernels' guard flagged (NULL: not recorded); repair.hip
};

StreamState *ann_stream_state(annchor_ctx *c, bool create);
// comm.hip: hipStreamSynchronize under the dead-peer watchdog; the engine stream's wait for a side-stream all-gather in flight
hipError_t ann_comm_guarded_sync(annchor_ctx *c, hipStream_t stream, const char *where);
int ann_comm_side_join(annchor_ctx *c);
// every host wait of the streamed form: guarded when the context holds a communicator (a dead peer must not block it for good)
inline hipError_t ann_stream_wait(annchor_ctx *c, const char *where)
{
    return c->comm ? ann_comm_guarded_sync(c, c->stream, where) : hipStreamSynchronize(c->stream);
}
// knnbf.hip: the tile phase on the 16-bit matrix cores (split operands, two 4-wave workgroups per CU); *handled = false
// when the shape does not fit it (padded dim > 128, more than 30 neighbours, no split copy) and the caller launches k_st_knn
int ann_stream_launch_knnbf(annchor_ctx *c, const struct KnnArgs &a, int dim_padded, bool *handled, bool join = false);
// tools/experiments/knnbf2.hip (builds with -DST_PAIR_KERNEL only): the same tile phase with two adjacent row tiles per 8-wave
// workgroup sharing one column stream (graph builds, padded dim <= 128, K + 2 <= 16); *handled = false otherwise
int ann_stream_launch_knnbf2(annchor_ctx *c, const struct KnnArgs &a, int dim_padded, bool *handled);
// knnh.hip: the two-stage tile kernel (fp16 hi-only products as a rigorous filter, float32 differences for what passes), warm-started by
// `warm` (k_st_knnbf with a small budget); graph builds at padded dimension 128, <= 14 neighbours; *handled = false otherwise
bool ann_stream_knnh_fits(const struct KnnArgs &a, int dim_padded);
int ann_stream_launch_knnh(annchor_ctx *c, const struct KnnArgs &a, int dim_padded, bool *handled,
                           int (*warm)(annchor_ctx *, const struct KnnArgs &, int, bool *, bool));
// knnbk.hip: the k-blocked split-fp16 kernel for padded dim 256 .. 1024 (tile phase and join passes, graph builds and queries)
int ann_stream_launch_knnbk(annchor_ctx *c, const struct KnnArgs &a, int dim_padded, bool *handled, bool join = false);
int ann_stream_split_rows(annchor_ctx *c, StreamState *s);     // Xb from Xs (after the ordering)
// repair.hip: the row tiles the split kernels' guard flagged, done again with float32 DIFFERENCES (exact; any dimension)
int ann_stream_repair_flagged(annchor_ctx *c, struct StreamState *s, const struct KnnArgs &a, int dim_padded, const uint32_t *guard_tiles, int64_t flagged);
// the split copy (+ centred norms, centre) that belongs to an ordered float32 array; false: none
bool ann_stream_split_of(const void *Xs, const uint16_t **Xb, const float **rsb, const float **cvec);
int ann_stream_reserve(annchor_ctx *c, DevBuf &b, size_t bytes);   // individual allocation, grow-only, contents NOT kept
int ann_stream_padded_dim(int dim);
// exact float32 distances of the kept neighbours, final order, ids: *d_idx_out int64 [rows][K], *d_dist_out float [rows][K]
int ann_stream_knn_finish(annchor_ctx *c, StreamState *s, KnnArgs &a, const void *perm_all, int dim_padded, int64_t **d_idx_out,
                          float **d_dist_out, int64_t *tile_evals);
void ann_stream_free_run(StreamState *s);
// distances of all bound rows to avec_dev (device, float [dim]) -> D[round][.], running minimum (reset for rounds 0 and 1),
// per-workgroup arg-max partials in red_val / red_idx; returns their count.  No host wait.
int ann_stream_sweep(annchor_ctx *c, StreamState *s, const float *avec_dev, int round, int *n_partials);
