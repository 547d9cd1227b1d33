// ifr_api.hip -- C ABI (include/ifr.h) over the kernels in ifr_kernels.h / ifr_aux.hip.
// Host side: owns device copies of the model and CSR data, picks the launch geometry,
// launches on the caller's stream.  No torch types; PyTorch only supplies device buffers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ifr.h"
#include "ifr_kernels.h"
#include "ifr_itemitem.h"
#include "ifr_fit.h"
#include "ifr_evalcsr.h"

namespace ifr {
// implemented in ifr_aux.hip
int launch_auc(const float *scores, int64_t ld, const int32_t *user_ids, int64_t n, int n_items,
               const int64_t *ex_ptr, const int32_t *ex_idx, int64_t ex_rows,
               const int64_t *tr_ptr, const int32_t *tr_idx, int64_t tr_rows, int max_pos,
               double *out, hipStream_t stream);
int launch_rank_items(const float *scores, int64_t n_scores, const int32_t *items, int64_t m, int32_t *out_ids,
                      void **ws, size_t *ws_bytes, hipStream_t stream);
int launch_select_topk(const float *scores, int64_t ld, const int32_t *user_ids, int64_t n, int n_items, const int64_t *ex_ptr,
                       const int32_t *ex_idx, int64_t ex_rows, int K, int32_t *out_ids, float *out_scores, hipStream_t stream);
int select_topk_max_k();
// fp64 score rows (item-item models scored in float64): the same consumers on 64-bit keys
int launch_auc_f64(const double *scores, int64_t ld, const int32_t *user_ids, int64_t n, int n_items,
                   const int64_t *ex_ptr, const int32_t *ex_idx, int64_t ex_rows,
                   const int64_t *tr_ptr, const int32_t *tr_idx, int64_t tr_rows, int max_pos, double *out, hipStream_t stream);
int auc_max_pos_f64();
int launch_rank_items_f64(const double *scores, int64_t n_scores, const int32_t *items, int64_t m, int32_t *out_ids,
                          void **ws, size_t *ws_bytes, hipStream_t stream);
int launch_select_topk_f64(const double *scores, int64_t ld, const int32_t *user_ids, int64_t n, int n_items, const int64_t *ex_ptr,
                           const int32_t *ex_idx, int64_t ex_rows, int K, int32_t *out_ids, float *out_scores, hipStream_t stream);
int launch_rank_full_f64(const double *scores, int64_t ld, const int32_t *user_ids, int64_t n, int n_items,
                         const int64_t *ex_ptr, const int32_t *ex_idx, int64_t ex_rows,
                         int32_t *out_ids, int32_t *n_cand, void **ws, size_t *ws_bytes, hipStream_t stream);
int launch_pad_rows_min_f64(double *scores, int64_t n, int64_t ld, int n_known, int n_cat, hipStream_t stream);
int launch_rank_full(const float *scores, int64_t ld, const int32_t *user_ids, int64_t n, int n_items,
                     const int64_t *ex_ptr, const int32_t *ex_idx, int64_t ex_rows,
                     int32_t *out_ids, int32_t *n_cand, void **ws, size_t *ws_bytes, hipStream_t stream);
}  // namespace ifr

enum { MODE_NONE = 0, MODE_FACTOR = 1, MODE_ITEM_DENSE = 2, MODE_ITEM_CSR = 3 };

// Workspace of ifr_rank_topk_noalloc (include/ifr.h): partial top-K lists + per-user bound words of one call.
// Every GPU the library runs on holds its own copy until the process ends; host code never uses the array's
// name as an address, only the per-device pointer hipGetSymbolAddress returned (scratch_of_device).
static __device__ __align__(256) unsigned char scratch[IFR_NOALLOC_SCRATCH_BYTES];
enum { MAX_DEVICES = 64 };
static void *g_scratch_ptr[MAX_DEVICES];

// address of `scratch` on the CURRENT device (must be `device`), looked up once per device
static void *scratch_of_device(int device)
{
    if (device < 0 || device >= MAX_DEVICES) return nullptr;
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess) return nullptr;
        g_scratch_ptr[device] = p;
    }
    return g_scratch_ptr[device];
}

struct ifr_handle_s {
    int device = 0;
    std::string err;
    // model
    float *U = nullptr, *V = nullptr, *ibase = nullptr, *ubase = nullptr;
    int64_t n_users = 0, n_items = 0;
    int k = 0, KD = 0, n_tiles = 0;
    bool has_base = false;
    int mode = MODE_NONE;
    // item-item model: dense M or chunk-blocked CSR, column scale, user histories
    float *iiM = nullptr; int64_t ii_ld = 0;
    double *iiM64 = nullptr;     // dense item matrix of the fp64 scoring option (then iiM is null)
    int opt_item_f64 = 0;        // ifr_fit_ease installs B in fp64
    int64_t *ii_cptr = nullptr; int32_t *ii_idx = nullptr; float *ii_val = nullptr; int64_t ii_nnz = 0;
    uint2 *ii_pair = nullptr;    // CSR entries as interleaved (column, value) pairs: what the CSR top-K kernel streams
    double ii_long_frac = 0.0;   // share of (row, 1024-column window) segments longer than that kernel's prefetch
    void *ii_win = nullptr; int ii_win_words = 1;      // 16-byte window words of the wave-per-user kernel (1 - 3 per row and window), or null
    uint32_t *ii_cptr_w = nullptr; int ii_n_win = 0;   // entry offsets per 1024-column window (wave-per-user CSR top-K kernel), or null
    void *ws_long = nullptr; size_t ws_long_bytes = 0;  // users that kernel hands over: [0] = count, then the list
    int opt_csr_wave = 1;        // 0: CSR top-K by the workgroup-per-user kernel only
    float *ii_pen = nullptr;
    int ii_chunks = 0;
    float ii_maxabs = 0.f;
    int64_t *hs_ptr = nullptr; int32_t *hs_idx = nullptr; float *hs_val = nullptr;
    int64_t hs_rows = 0, hs_nnz = 0;
    float hs_l1max = 0.f;
    double prof_bytes = 0.0;
    size_t dbg_stamp_count = 0;
    // exclusions: raw CSR (AUC / full rank) + tile-encoded lists (fused kernel)
    int64_t *ex_ptr = nullptr;
    int32_t *ex_idx = nullptr, *ex_cnt = nullptr;
    uint2 *ex_pair = nullptr;
    int64_t ex_rows = 0, ex_nnz = 0;
    // truth
    int64_t *tr_ptr = nullptr;
    int32_t *tr_idx = nullptr;
    int64_t tr_rows = 0, tr_nnz = 0;
    int tr_max_row = 0;
    // workspaces (grow-only)
    void *rocblas = nullptr;     // rocBLAS handle of the model build entry points (created on first use)
    void *rocsparse = nullptr;   // rocSPARSE handle, likewise
    void *ws_part = nullptr; size_t ws_part_bytes = 0;
    void *ws_met = nullptr; size_t ws_met_bytes = 0;
    void *ws_sort = nullptr; size_t ws_sort_bytes = 0;
    double *discount = nullptr;
    void *scratch = nullptr;     // this device's copy of the static `scratch` array (ifr_rank_topk_noalloc)
    // tuning
    int opt_chunks = 0;          // 0 = heuristic
    int opt_target_blocks = 4096;
    int opt_share_bounds = 1;    // share per-user lower bounds between item chunks
    int opt_wide_stats = 0;      // count the candidates of the 16 < K <= 128 path (two contended atomics per user: measurement runs only)
    int opt_wide_chunk_cap = 2;  // item chunks of that path beyond the ceil(K / 24) its bound needs (more lists loosen the bound)
    int opt_tail_split = 1;      // two-region grid: the last, partial round of workgroups runs shorter item chunks
    int opt_pair_div = 12;       // list path (16 < K): one item chunk (two 16-entry lists per user) per pair_div wanted items
    int opt_pair_max = 100;      // largest K of that path (at most 128).  Measured against the candidate buffers: K = 50 -3 % (C2) / -10 %
                                 // (C3 shape) per pass, K = 100 -11.5 % / -5.6 % (0.7 % of the users are redone, at C3 shape from 4 MB
                                 // rows); K = 128 fills half of the 256 list slots and is redone for 9 % of the users.  With contiguous
                                 // item chunks a catalogue sorted by score puts a whole top K into one chunk and every user is redone:
                                 // pair_limit below then turns the path off for that model
    int pair_limit = IFR_MAX_TOPK;   // adaptive: a call of the pair path that had to redo more than 5 % of its users (a catalogue sorted
                                 // by score puts a user's whole top K into one list) sends that K and larger ones to the
                                 // candidate-buffer path from then on; reset with the model
    int opt_pair_topk = 1;       // 16 < K <= 32 of a dot-product model: two 16-lists per user and item chunk + loss check at the
                                 // merge (0: the candidate-buffer path like K > 32)
    int opt_debug = 0;           // dev-only kernel ablation mode (see TopkParams::dbg)
    int opt_csr_pass = 0;        // test: candidate buffer entries of the CSR top-K kernel (0 = all; small values force its exact argmax fallback)
    int64_t opt_catalogue = 0;   // items the dense-row consumers rank (0 = n_items; > n_items: items unknown to the model)
    // profiling
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_used, ev_free;
    std::vector<std::pair<const void *, int>> slots_cache;     // fused kernel -> resident workgroups on the chip
    double prof_pairs = 0.0;
    const uint32_t *dbg_wide_cnt = nullptr; int64_t dbg_wide_n = 0;     // dev: candidate counts of the last 16 < K <= 128 call
    int64_t wide_users = 0;      // ... users that path ranked, candidates it appended / kept after the final bound
    double wide_cand = 0.0, wide_kept = 0.0;
    int64_t wide_overflows = 0;  // users of the 16 < K <= 128 path ranked from dense rows because their candidate buffer overflowed
};

namespace {

int fail(ifr_handle h, int code, const std::string &msg)
{
    if (h) h->err = msg;
    return code;
}

#define IFR_HIP(h, call)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(h, e_ == hipErrorOutOfMemory ? IFR_ERR_NOMEM : IFR_ERR_HIP,            \
                        std::string(#call) + ": " + hipGetErrorString(e_));                   \
    } while (0)

template <typename T>
void dfree(T *&p)
{
    if (p) { (void)hipFree(p); p = nullptr; }
}

int grow(ifr_handle h, void **p, size_t *have, size_t need)
{
    if (*have >= need) return IFR_OK;
    if (*p) { (void)hipFree(*p); *p = nullptr; *have = 0; }
    IFR_HIP(h, hipMalloc(p, need));
    *have = need;
    return IFR_OK;
}

// factor dimension of the kernel instantiation: steps of 32 up to 256 (the reference's WMF default k = 200 runs
// as 224, k = 100 as 128, k = 50 as 64)
int pad_k(int k) { return k <= 32 ? 32 : (k + 31) / 32 * 32; }

// cap_chunks > 0: an upper limit on the item chunks (a fixed-size workspace holds 2 partial lists per chunk and user)
void pick_chunks(ifr_handle h, int64_t n, int *chunk_tiles, int *n_chunks, int64_t cap_chunks = 0)
{
    const int64_t n_ut = (n + ifr::USERS_PER_BLOCK - 1) / ifr::USERS_PER_BLOCK;
    int64_t c = h->opt_chunks > 0 ? h->opt_chunks : (h->opt_target_blocks + n_ut - 1) / n_ut;
    const int64_t max_c = std::max<int64_t>(1, std::min<int64_t>(64, h->n_tiles / 16));
    c = std::max<int64_t>(1, std::min(c, max_c));
    if (cap_chunks > 0) c = std::min(c, cap_chunks);
    *chunk_tiles = (int)((h->n_tiles + c - 1) / c);
    *n_chunks = (h->n_tiles + *chunk_tiles - 1) / *chunk_tiles;
}

typedef void (*topk_fn)(ifr::TopkParams);

// the fused kernel's instantiation: lane lists of 16 entries (K <= 16 exact; WIDE: 16 < K <= 128 through candidate buffers)
template <int KD>
topk_fn topk_kernel(bool has_base, bool store, bool wide)
{
    if (wide) {
        if (store) return has_base ? ifr::score_topk_kernel<KD, 16, true, true, true> : ifr::score_topk_kernel<KD, 16, false, true, true>;
        return has_base ? ifr::score_topk_kernel<KD, 16, true, false, true> : ifr::score_topk_kernel<KD, 16, false, false, true>;
    }
    if (store) return has_base ? ifr::score_topk_kernel<KD, 16, true, true> : ifr::score_topk_kernel<KD, 16, false, true>;
    return has_base ? ifr::score_topk_kernel<KD, 16, true> : ifr::score_topk_kernel<KD, 16, false>;
}

topk_fn topk_kernel_kd(int KD, bool has_base, bool store, bool wide)
{
    if (KD == 32) return topk_kernel<32>(has_base, store, wide);
    if (KD == 64) return topk_kernel<64>(has_base, store, wide);
    if (KD == 96) return topk_kernel<96>(has_base, store, wide);
    if (KD == 128) return topk_kernel<128>(has_base, store, wide);
    if (KD == 160) return topk_kernel<160>(has_base, store, wide);
    if (KD == 192) return topk_kernel<192>(has_base, store, wide);
    if (KD == 224) return topk_kernel<224>(has_base, store, wide);
    return topk_kernel<256>(has_base, store, wide);
}

// workgroups of `fn` that are resident on the whole chip at once (occupancy API x CUs; cached per kernel)
int resident_slots(ifr_handle h, topk_fn fn, size_t dyn_lds)
{
    const void *key = reinterpret_cast<const void *>(fn);
    for (auto &e : h->slots_cache)
        if (e.first == key) return e.second;
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, key, ifr::BLOCK, dyn_lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || cus < 1) cus = 256;
    h->slots_cache.push_back({key, per_cu * cus});
    return per_cu * cus;
}

// Launch plan of the fused kernel: W = n_ut * c equal workgroups on S resident slots run in ceil(W / S) rounds, and
// the last round keeps only (W mod S) slots busy for a whole workgroup's time.  Region A = the user tiles that fill
// whole rounds with c item chunks each; the rest (region B) is cut into more, shorter chunks that fill ONE round
// together.  c itself is chosen by a cost model in item-tile times: rounds x (tiles per chunk + a fixed cost per
// workgroup -- its first tiles fill the lists at full insertion cost, measured ~3 tile times: C2 with one chunk per
// user tile runs 0.9 % faster than with four) instead of a fixed workgroup count.
struct GridPlan { int chunk_tiles, n_chunks, n_utA, chunk_tilesB, n_chunksB; };

GridPlan plan_grid(ifr_handle h, int64_t n_ut, int64_t S, int min_chunks, int64_t cap_chunks)
{
    const int64_t T = h->n_tiles;
    // both limits apply: at most 64 item chunks (the merge reads 2 lists per chunk from LDS) AND the caller's cap (a
    // fixed-size workspace: a small batch leaves room for hundreds of chunks, which the merge could not take)
    const int64_t max_c = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(64, cap_chunks > 0 ? cap_chunks : 64), T / 16));
    auto region = [&](int64_t c, int *tiles, int *chunks) {
        *tiles = (int)((T + c - 1) / c);
        *chunks = (int)((T + *tiles - 1) / *tiles);
    };
    // the list path wants min_chunks list pairs per user even on a catalogue of a few tiles (one region then)
    auto at_least = [&](GridPlan g) {
        if (g.n_chunks >= min_chunks || T < 2) return g;
        g.chunk_tiles = (int)std::max<int64_t>(1, T / min_chunks);
        g.n_chunks = (int)((T + g.chunk_tiles - 1) / g.chunk_tiles);
        if (cap_chunks > 0 && g.n_chunks > cap_chunks) region(cap_chunks, &g.chunk_tiles, &g.n_chunks);
        g.n_utA = (int)n_ut; g.chunk_tilesB = g.chunk_tiles; g.n_chunksB = g.n_chunks;
        return g;
    };
    GridPlan best{};
    if (h->opt_chunks > 0 || !h->opt_tail_split) {                 // forced chunk count, or the one-region heuristic
        int64_t c = h->opt_chunks > 0 ? h->opt_chunks : (h->opt_target_blocks + n_ut - 1) / n_ut;
        c = std::max<int64_t>(1, std::min(c, std::max<int64_t>(1, std::min<int64_t>(64, T / 16))));
        if (cap_chunks > 0) c = std::min(c, cap_chunks);
        c = std::max<int64_t>(c, std::min<int64_t>(min_chunks, std::max<int64_t>(1, std::min<int64_t>(T, 8))));
        region(c, &best.chunk_tiles, &best.n_chunks);
        best.n_utA = (int)n_ut; best.chunk_tilesB = best.chunk_tiles; best.n_chunksB = best.n_chunks;
        return at_least(best);
    }
    const double wg_cost = 3.0;
    const int64_t c_lo = std::min<int64_t>(std::max<int64_t>(1, min_chunks), std::max<int64_t>(1, std::min<int64_t>(T, 8)));
    const int64_t c_hi = std::max(c_lo, std::min(max_c, (h->opt_target_blocks + n_ut - 1) / n_ut));
    double best_t = 0;
    for (int64_t c = c_lo; c <= c_hi; c++) {
        GridPlan g;
        region(c, &g.chunk_tiles, &g.n_chunks);
        g.n_utA = (int)n_ut; g.chunk_tilesB = g.chunk_tiles; g.n_chunksB = g.n_chunks;
        const int64_t W = n_ut * g.n_chunks, R = W / S, f = W - R * S;
        double t = (double)((W + S - 1) / S) * (g.chunk_tiles + wg_cost);
        if (R >= 1 && f > 0 && 4 * f < 3 * S) {
            const int64_t a = R * S / g.n_chunks, b = n_ut - a;                  // user tiles of regions A and B
            const int64_t cB = std::min(max_c, S / std::max<int64_t>(b, 1));
            if (b > 0 && a > 0 && cB > g.n_chunks) {
                g.n_utA = (int)a;
                region(cB, &g.chunk_tilesB, &g.n_chunksB);
                t = (double)R * (g.chunk_tiles + wg_cost) + (g.chunk_tilesB + wg_cost);
            }
        }
        if (c == c_lo || t < best_t) { best = g; best_t = t; }
    }
    return at_least(best);
}

template <int KD>
void launch_dense(bool has_base, const ifr::DenseParams &p, int grid, hipStream_t s)
{
    if (has_base) hipLaunchKernelGGL((ifr::score_dense_kernel<KD, true>), dim3(grid), dim3(ifr::BLOCK), 0, s, p);
    else hipLaunchKernelGGL((ifr::score_dense_kernel<KD, false>), dim3(grid), dim3(ifr::BLOCK), 0, s, p);
}

int64_t catalogue_items(ifr_handle h) { return h->opt_catalogue > h->n_items ? h->opt_catalogue : h->n_items; }

void clear_item_model(ifr_handle h)
{
    dfree(h->iiM); dfree(h->iiM64); dfree(h->ii_cptr); dfree(h->ii_idx); dfree(h->ii_val); dfree(h->ii_pen); dfree(h->ii_pair); dfree(h->ii_cptr_w); dfree(h->ii_win); h->ii_n_win = 0;
    dfree(h->hs_ptr); dfree(h->hs_idx); dfree(h->hs_val);
    h->ii_ld = 0; h->ii_nnz = 0; h->ii_chunks = 0; h->ii_maxabs = 0.f;
    h->hs_rows = 0; h->hs_nnz = 0; h->hs_l1max = 0.f;
    if (h->mode == MODE_ITEM_DENSE || h->mode == MODE_ITEM_CSR) { h->mode = MODE_NONE; h->n_items = 0; }
}

void clear_factor_model(ifr_handle h)
{
    dfree(h->U); dfree(h->V); dfree(h->ibase); dfree(h->ubase);
    h->n_users = 0; h->k = 0; h->KD = 0; h->n_tiles = 0; h->has_base = false;
    h->pair_limit = IFR_MAX_TOPK;
    if (h->mode == MODE_FACTOR) { h->mode = MODE_NONE; h->n_items = 0; }
}

// CSR data was laid out for another catalogue: drop it (exclusions AND truth) rather than index out of range later
void drop_csrs_if_catalogue_changes(ifr_handle h, int64_t n_items)
{
    if (h->n_items == n_items) return;
    dfree(h->ex_ptr); dfree(h->ex_idx); dfree(h->ex_cnt); dfree(h->ex_pair);
    h->ex_rows = 0; h->ex_nnz = 0;
    dfree(h->tr_ptr); dfree(h->tr_idx);
    h->tr_rows = 0; h->tr_nnz = 0; h->tr_max_row = 0;
}

// column scale padded to whole chunks (the kernels read it 16 B at a time)
int set_col_scale(ifr_handle h, const float *col_scale, int64_t n_items, int64_t padded)
{
    dfree(h->ii_pen);
    if (!col_scale) return IFR_OK;
    IFR_HIP(h, hipMalloc((void **)&h->ii_pen, sizeof(float) * padded));
    IFR_HIP(h, hipMemset(h->ii_pen, 0, sizeof(float) * padded));
    IFR_HIP(h, hipMemcpy(h->ii_pen, col_scale, sizeof(float) * n_items, hipMemcpyDefault));
    return IFR_OK;
}

// 2^shift: the largest power of two for which sum_j |x_uj M[j,c]| * 2^shift <= 2^51 for every user: every product and
// every partial sum then is an integer below 2^51 in magnitude -- exact in int64 AND in fp64, whose adds therefore do not
// round (the top-K kernel accumulates integer-valued doubles with ds_add_f64; round 2 used 2^61 and int64 only)
int fixed_point_shift(ifr_handle h)
{
    const double bound = (double)h->ii_maxabs * (double)h->hs_l1max * 1.001;
    if (!(bound > 0.0)) return 0;
    int e = 0;
    (void)std::frexp(bound, &e);                 // bound <= 2^e
    return std::max(-960, std::min(960, 51 - e));
}

int read_flag_and_max(ifr_handle h, const void *dev, uint32_t *out)
{
    IFR_HIP(h, hipMemcpy(out, dev, sizeof(uint32_t), hipMemcpyDeviceToHost));
    return IFR_OK;
}

}  // namespace

extern "C" {

const char *ifr_version(void) { return "ifr 0.1 (gfx950)"; }

int ifr_create(ifr_handle *out, int device_id)
{
    if (!out) return IFR_ERR_ARG;
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device_id < 0 || device_id >= n_dev) return IFR_ERR_HIP;
    if (hipSetDevice(device_id) != hipSuccess) return IFR_ERR_HIP;
    ifr_handle h = new ifr_handle_s();
    h->device = device_id;
    double tab[IFR_MAX_TOPK];
    for (int i = 0; i < IFR_MAX_TOPK; ++i) tab[i] = 1.0 / std::log2((double)i + 2.0);
    if (hipMalloc((void **)&h->discount, sizeof(tab)) != hipSuccess ||
        hipMemcpy(h->discount, tab, sizeof(tab), hipMemcpyHostToDevice) != hipSuccess) {
        delete h;
        return IFR_ERR_HIP;
    }
    h->scratch = scratch_of_device(device_id);
    if (!h->scratch) {
        (void)hipFree(h->discount);
        delete h;
        return IFR_ERR_HIP;
    }
    *out = h;
    return IFR_OK;
}

int ifr_destroy(ifr_handle h)
{
    if (!h) return IFR_ERR_ARG;
    (void)hipSetDevice(h->device);
    dfree(h->U); dfree(h->V); dfree(h->ibase); dfree(h->ubase);
    dfree(h->ex_ptr); dfree(h->ex_idx); dfree(h->ex_cnt); dfree(h->ex_pair);
    dfree(h->tr_ptr); dfree(h->tr_idx); dfree(h->discount);
    dfree(h->iiM); dfree(h->iiM64); dfree(h->ii_cptr); dfree(h->ii_idx); dfree(h->ii_val); dfree(h->ii_pen); dfree(h->ii_pair); dfree(h->ii_cptr_w); dfree(h->ii_win); h->ii_n_win = 0;
    dfree(h->hs_ptr); dfree(h->hs_idx); dfree(h->hs_val);
    dfree(h->ws_part); dfree(h->ws_met); dfree(h->ws_sort); dfree(h->ws_long);
    ifr::ease_release(&h->rocblas);
    ifr::rp3_release(&h->rocsparse);
    for (auto &e : h->ev_used) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto &e : h->ev_free) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    delete h;
    return IFR_OK;
}

const char *ifr_last_error(ifr_handle h) { return h ? h->err.c_str() : "null handle"; }

int ifr_set_option(ifr_handle h, const char *name, int64_t value)
{
    if (!h || !name) return IFR_ERR_ARG;
    if (!strcmp(name, "chunks")) h->opt_chunks = (int)value;
    else if (!strcmp(name, "share_bounds")) h->opt_share_bounds = (int)value;
    else if (!strcmp(name, "tail_split")) h->opt_tail_split = (int)value;
    else if (!strcmp(name, "pair_topk")) h->opt_pair_topk = value != 0;
    else if (!strcmp(name, "pair_div")) h->opt_pair_div = (int)std::max<int64_t>(4, std::min<int64_t>(value, 64));
    else if (!strcmp(name, "pair_max")) h->opt_pair_max = (int)std::max<int64_t>(16, std::min<int64_t>(value, IFR_MAX_TOPK));
    else if (!strcmp(name, "item_f64")) h->opt_item_f64 = value != 0;
    else if (!strcmp(name, "wide_stats")) h->opt_wide_stats = (int)value;
    else if (!strcmp(name, "wide_chunk_cap")) h->opt_wide_chunk_cap = (int)std::max<int64_t>(1, value);
    else if (!strcmp(name, "debug_mode")) {
#if !defined(IFR_ABLATE) && !defined(IFR_STAMPS)
        // bits 0-2 (kernel ablations of the fused kernel) and bit 7 (phase stamps) only exist in a dev build
        if (value & (7 | 128)) return fail(h, IFR_ERR_ARG, "ifr_set_option: debug_mode ablations need a library built with make ABLATE=1 / STAMPS=1");
#endif
        h->opt_debug = (int)value;
    }
    else if (!strcmp(name, "csr_wave")) h->opt_csr_wave = value <= 0 ? 0 : (value == 1 ? 1 : 2);
    else if (!strcmp(name, "csr_cand_cap")) h->opt_csr_pass = value <= 0 ? 0 : (int)std::max<int64_t>(ifr::item_sparse_min_capacity(), std::min<int64_t>(value, ifr::item_sparse_pass_capacity()));
    else if (!strcmp(name, "catalogue_items")) {
        if (value != 0 && (value < h->n_items || value > (int64_t)1 << 30))
            return fail(h, IFR_ERR_ARG, "catalogue_items must be 0 or >= the model's n_items");
        h->opt_catalogue = value;
    }
    else if (!strcmp(name, "target_blocks")) h->opt_target_blocks = (int)std::max<int64_t>(1, value);
    else return fail(h, IFR_ERR_ARG, std::string("unknown option ") + name);
    return IFR_OK;
}

int ifr_set_factors(ifr_handle h, const float *U, int64_t n_users, const float *V, int64_t n_items,
                    int k, const float *ubase, const float *ibase)
{
    if (!h) return IFR_ERR_ARG;
    if (!U || !V || n_users <= 0 || n_items <= 0 || k <= 0) return fail(h, IFR_ERR_ARG, "ifr_set_factors: null/empty input");
    if (k > IFR_MAX_FACTORS) return fail(h, IFR_ERR_ARG, "ifr_set_factors: k > IFR_MAX_FACTORS (256)");
    if (n_items > (int64_t)1 << 30) return fail(h, IFR_ERR_ARG, "ifr_set_factors: n_items too large");
    IFR_HIP(h, hipSetDevice(h->device));
    const int64_t old_items = h->n_items;
    clear_factor_model(h);                 // mode NONE, no users / items: a failed allocation below leaves no model set
    clear_item_model(h);
    h->n_items = old_items;                // only for the catalogue comparison below
    drop_csrs_if_catalogue_changes(h, n_items);
    h->n_items = 0;
    const int KD = pad_k(k);
    const int n_tiles = (int)((n_items + ifr::TILE_I - 1) / ifr::TILE_I);
    const int64_t v_rows = (int64_t)(n_tiles + 1) * ifr::TILE_I;
    IFR_HIP(h, hipMalloc((void **)&h->U, sizeof(float) * n_users * KD));
    IFR_HIP(h, hipMalloc((void **)&h->V, sizeof(float) * v_rows * KD));
    IFR_HIP(h, hipMalloc((void **)&h->ibase, sizeof(float) * v_rows));
    IFR_HIP(h, hipMalloc((void **)&h->ubase, sizeof(float) * n_users));
    IFR_HIP(h, hipMemset(h->U, 0, sizeof(float) * n_users * KD));
    IFR_HIP(h, hipMemset(h->V, 0, sizeof(float) * v_rows * KD));
    IFR_HIP(h, hipMemset(h->ibase, 0, sizeof(float) * v_rows));
    IFR_HIP(h, hipMemset(h->ubase, 0, sizeof(float) * n_users));
    IFR_HIP(h, hipMemcpy2D(h->U, sizeof(float) * KD, U, sizeof(float) * k, sizeof(float) * k, n_users, hipMemcpyDefault));
    IFR_HIP(h, hipMemcpy2D(h->V, sizeof(float) * KD, V, sizeof(float) * k, sizeof(float) * k, n_items, hipMemcpyDefault));
    if (ibase) IFR_HIP(h, hipMemcpy(h->ibase, ibase, sizeof(float) * n_items, hipMemcpyDefault));
    if (ubase) IFR_HIP(h, hipMemcpy(h->ubase, ubase, sizeof(float) * n_users, hipMemcpyDefault));
    h->has_base = (ibase != nullptr) || (ubase != nullptr);
    h->n_users = n_users; h->n_items = n_items; h->k = k; h->KD = KD; h->n_tiles = n_tiles;
    h->mode = MODE_FACTOR;
    h->opt_catalogue = 0;
    return IFR_OK;
}

int ifr_set_user_factors(ifr_handle h, const float *U, int64_t n_users, int k, const float *ubase)
{
    if (!h) return IFR_ERR_ARG;
    if (h->mode != MODE_FACTOR) return fail(h, IFR_ERR_STATE, "ifr_set_user_factors: call ifr_set_factors first");
    if (!U || n_users <= 0 || k != h->k) return fail(h, IFR_ERR_ARG, "ifr_set_user_factors: null/empty input or k differs from the item factors'");
    if (ubase && !h->has_base) return fail(h, IFR_ERR_ARG, "ifr_set_user_factors: the model was set without bases");
    IFR_HIP(h, hipSetDevice(h->device));
    dfree(h->U); dfree(h->ubase);
    h->n_users = 0;
    const int KD = h->KD;
    IFR_HIP(h, hipMalloc((void **)&h->U, sizeof(float) * n_users * KD));
    IFR_HIP(h, hipMalloc((void **)&h->ubase, sizeof(float) * n_users));
    IFR_HIP(h, hipMemset(h->U, 0, sizeof(float) * n_users * KD));
    IFR_HIP(h, hipMemset(h->ubase, 0, sizeof(float) * n_users));
    IFR_HIP(h, hipMemcpy2D(h->U, sizeof(float) * KD, U, sizeof(float) * k, sizeof(float) * k, n_users, hipMemcpyDefault));
    if (ubase) IFR_HIP(h, hipMemcpy(h->ubase, ubase, sizeof(float) * n_users, hipMemcpyDefault));
    h->n_users = n_users;
    return IFR_OK;
}

static int copy_csr(ifr_handle h, const int64_t *indptr, const int32_t *indices, int64_t n_rows, int64_t nnz,
                    int64_t **d_ptr, int32_t **d_idx, std::vector<int64_t> *host_ptr)
{
    dfree(*d_ptr); dfree(*d_idx);
    host_ptr->assign(n_rows + 1, 0);
    IFR_HIP(h, hipMemcpy(host_ptr->data(), indptr, sizeof(int64_t) * (n_rows + 1), hipMemcpyDefault));
    if ((*host_ptr)[0] != 0 || (*host_ptr)[n_rows] != nnz) return fail(h, IFR_ERR_ARG, "CSR: indptr[0] != 0 or indptr[n_rows] != nnz");
    for (int64_t r = 0; r < n_rows; ++r)
        if ((*host_ptr)[r + 1] < (*host_ptr)[r]) return fail(h, IFR_ERR_ARG, "CSR: indptr not monotone");
    IFR_HIP(h, hipMalloc((void **)d_ptr, sizeof(int64_t) * (n_rows + 1)));
    IFR_HIP(h, hipMalloc((void **)d_idx, sizeof(int32_t) * (nnz + ifr::item_csr_pad())));     // padded: see item_csr_pad
    IFR_HIP(h, hipMemcpy(*d_ptr, host_ptr->data(), sizeof(int64_t) * (n_rows + 1), hipMemcpyHostToDevice));
    if (nnz > 0) IFR_HIP(h, hipMemcpy(*d_idx, indices, sizeof(int32_t) * nnz, hipMemcpyDefault));
    return IFR_OK;
}

// h->ex_ptr / h->ex_idx are in place: the per-tile encoding the fused kernel reads
static int encode_exclusions(ifr_handle h, int64_t n_rows, int64_t nnz)
{
    IFR_HIP(h, hipMalloc((void **)&h->ex_cnt, sizeof(int32_t) * n_rows));
    IFR_HIP(h, hipMalloc((void **)&h->ex_pair, sizeof(uint2) * std::max<int64_t>(nnz, 1)));
    const int grid = (int)((n_rows + 255) / 256);
    hipLaunchKernelGGL(ifr::encode_exclusions_kernel, dim3(grid), dim3(256), 0, 0, h->ex_ptr, h->ex_idx, n_rows,
                       (int)h->n_items, h->ex_cnt, h->ex_pair);
    IFR_HIP(h, hipGetLastError());
    IFR_HIP(h, hipDeviceSynchronize());
    h->ex_rows = n_rows; h->ex_nnz = nnz;
    return IFR_OK;
}

int ifr_set_exclusion_csr(ifr_handle h, const int64_t *indptr, const int32_t *indices, int64_t n_rows, int64_t nnz)
{
    if (!h) return IFR_ERR_ARG;
    IFR_HIP(h, hipSetDevice(h->device));
    dfree(h->ex_ptr); dfree(h->ex_idx); dfree(h->ex_cnt); dfree(h->ex_pair);
    h->ex_rows = 0; h->ex_nnz = 0;
    if (n_rows <= 0 || !indptr) return IFR_OK;
    if (nnz > 0 && !indices) return fail(h, IFR_ERR_ARG, "ifr_set_exclusion_csr: null indices");
    if (h->n_items <= 0) return fail(h, IFR_ERR_STATE, "ifr_set_exclusion_csr: call ifr_set_factors first");
    std::vector<int64_t> hp;
    int rc = copy_csr(h, indptr, indices, n_rows, nnz, &h->ex_ptr, &h->ex_idx, &hp);
    if (rc) return rc;
    return encode_exclusions(h, n_rows, nnz);
}

int ifr_set_truth_csr(ifr_handle h, const int64_t *indptr, const int32_t *indices, int64_t n_rows, int64_t nnz)
{
    if (!h) return IFR_ERR_ARG;
    IFR_HIP(h, hipSetDevice(h->device));
    dfree(h->tr_ptr); dfree(h->tr_idx);
    h->tr_rows = 0; h->tr_nnz = 0; h->tr_max_row = 0;
    if (n_rows <= 0 || !indptr) return IFR_OK;
    if (nnz > 0 && !indices) return fail(h, IFR_ERR_ARG, "ifr_set_truth_csr: null indices");
    std::vector<int64_t> hp;
    int rc = copy_csr(h, indptr, indices, n_rows, nnz, &h->tr_ptr, &h->tr_idx, &hp);
    if (rc) return rc;
    int64_t mx = 0;
    for (int64_t r = 0; r < n_rows; ++r) mx = std::max(mx, hp[r + 1] - hp[r]);
    h->tr_rows = n_rows; h->tr_nnz = nnz; h->tr_max_row = (int)mx;
    return IFR_OK;
}

// The candidate-set builder of ranking_eval on the device (ifr_evalcsr.hip): the three raw rating matrices -> the truth
// and exclusion CSRs of user rows [row0, row0 + n_rows), installed in the handle like the two setters above would.
int ifr_build_eval_csrs(ifr_handle h, const ifr_csr_split *train, const ifr_csr_split *val, const ifr_csr_split *test,
                        int ratings_f64, double rating_threshold, int64_t n_items, int64_t row0, int64_t n_rows,
                        uint8_t *has_pos, int64_t *truth_nnz, int64_t *excl_nnz, int *max_truth_row)
{
    if (!h) return IFR_ERR_ARG;
    if (!train || !test || !train->indptr || !test->indptr) return fail(h, IFR_ERR_ARG, "ifr_build_eval_csrs: train and test splits are required");
    if (n_items <= 0 || n_items > (int64_t)1 << 30 || row0 < 0 || n_rows < 0) return fail(h, IFR_ERR_ARG, "ifr_build_eval_csrs: bad n_items / row range");
    if (h->n_items <= 0) return fail(h, IFR_ERR_STATE, "ifr_build_eval_csrs: set the model first");
    IFR_HIP(h, hipSetDevice(h->device));
    dfree(h->ex_ptr); dfree(h->ex_idx); dfree(h->ex_cnt); dfree(h->ex_pair);
    dfree(h->tr_ptr); dfree(h->tr_idx);
    h->ex_rows = 0; h->ex_nnz = 0; h->tr_rows = 0; h->tr_nnz = 0; h->tr_max_row = 0;
    if (truth_nnz) *truth_nnz = 0;
    if (excl_nnz) *excl_nnz = 0;
    if (max_truth_row) *max_truth_row = 0;
    if (n_rows == 0) return IFR_OK;

    ifr::EvalCsrParams p{};
    auto raw = [&](const ifr_csr_split *sp) {
        ifr::RawCsr r{};
        if (sp && sp->indptr && sp->n_rows > 0) { r.ptr = sp->indptr; r.idx = sp->indices; r.val = sp->ratings; r.rows = sp->n_rows; r.f64 = ratings_f64; }
        return r;
    };
    p.train = raw(train); p.val = raw(val); p.test = raw(test);
    p.thr = rating_threshold; p.n_items = (int)n_items; p.n_rows = n_rows; p.row0 = row0;
    uint8_t *d_has = nullptr;
    int *d_max = nullptr;
    int32_t *d_run = nullptr, *d_tmp = nullptr;
    void *d_scan = nullptr;
    const size_t scan_bytes = ifr::eval_csr_scan_bytes(n_rows);
    auto cleanup = [&]() { dfree(d_has); dfree(d_max); dfree(d_run); dfree(d_tmp); dfree(d_scan); };
    bool ok = hipMalloc((void **)&h->tr_ptr, sizeof(int64_t) * (n_rows + 1)) == hipSuccess &&
              hipMalloc((void **)&h->ex_ptr, sizeof(int64_t) * (n_rows + 1)) == hipSuccess &&
              hipMalloc((void **)&d_has, (size_t)n_rows) == hipSuccess && hipMalloc((void **)&d_max, sizeof(int)) == hipSuccess &&
              hipMalloc(&d_scan, std::max<size_t>(scan_bytes, 16)) == hipSuccess &&
              (p.val.ptr == nullptr || hipMalloc((void **)&d_run, sizeof(int32_t) * n_rows) == hipSuccess);
    if (!ok) { cleanup(); return fail(h, IFR_ERR_HIP, "ifr_build_eval_csrs: out of device memory (row pointers)"); }
    p.cnt_truth = h->tr_ptr; p.cnt_excl = h->ex_ptr; p.run_a = d_run; p.has_pos = d_has; p.max_truth_row = d_max;
    if (ifr::launch_eval_csr_count(p, d_scan, scan_bytes, 0) != 0) { cleanup(); return fail(h, IFR_ERR_HIP, "ifr_build_eval_csrs: count sweep failed"); }
    int64_t tn = 0, en = 0;
    int mx = 0;
    ok = hipMemcpy(&tn, h->tr_ptr + n_rows, sizeof(int64_t), hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(&en, h->ex_ptr + n_rows, sizeof(int64_t), hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(&mx, d_max, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    ok = ok && hipMalloc((void **)&h->tr_idx, sizeof(int32_t) * (tn + ifr::item_csr_pad())) == hipSuccess &&
         hipMalloc((void **)&h->ex_idx, sizeof(int32_t) * (en + ifr::item_csr_pad())) == hipSuccess &&
         (p.val.ptr == nullptr || hipMalloc((void **)&d_tmp, sizeof(int32_t) * std::max<int64_t>(en, 1)) == hipSuccess);
    if (!ok) { cleanup(); return fail(h, IFR_ERR_HIP, "ifr_build_eval_csrs: out of device memory (entries)"); }
    p.tr_ptr = h->tr_ptr; p.ex_ptr = h->ex_ptr; p.tr_idx = h->tr_idx; p.ex_idx = h->ex_idx; p.ex_tmp = d_tmp;
    if (ifr::launch_eval_csr_fill(p, 0) != 0) { cleanup(); return fail(h, IFR_ERR_HIP, "ifr_build_eval_csrs: fill sweep failed"); }
    if (has_pos && hipMemcpy(has_pos, d_has, (size_t)n_rows, hipMemcpyDefault) != hipSuccess) { cleanup(); return fail(h, IFR_ERR_HIP, "ifr_build_eval_csrs: has_pos copy failed"); }
    if (hipDeviceSynchronize() != hipSuccess) { cleanup(); return fail(h, IFR_ERR_HIP, "ifr_build_eval_csrs: kernels failed"); }
    cleanup();
    h->tr_rows = n_rows; h->tr_nnz = tn; h->tr_max_row = mx;
    if (truth_nnz) *truth_nnz = tn;
    if (excl_nnz) *excl_nnz = en;
    if (max_truth_row) *max_truth_row = mx;
    return encode_exclusions(h, n_rows, en);
}

int ifr_read_eval_csr(ifr_handle h, int which, int64_t *indptr, int32_t *indices, int64_t *n_rows, int64_t *nnz)
{
    if (!h || (which != 0 && which != 1)) return IFR_ERR_ARG;
    const int64_t rows = which == 0 ? h->tr_rows : h->ex_rows, n = which == 0 ? h->tr_nnz : h->ex_nnz;
    const int64_t *dp = which == 0 ? h->tr_ptr : h->ex_ptr;
    const int32_t *di = which == 0 ? h->tr_idx : h->ex_idx;
    if (n_rows) *n_rows = rows;
    if (nnz) *nnz = n;
    IFR_HIP(h, hipSetDevice(h->device));
    if (indptr && rows > 0) IFR_HIP(h, hipMemcpy(indptr, dp, sizeof(int64_t) * (rows + 1), hipMemcpyDefault));
    if (indices && n > 0) IFR_HIP(h, hipMemcpy(indices, di, sizeof(int32_t) * n, hipMemcpyDefault));
    return IFR_OK;
}


int ifr_set_item_matrix_dense(ifr_handle h, const float *M, int64_t n_items, int64_t ld, const float *col_scale)
{
    if (!h) return IFR_ERR_ARG;
    if (!M || n_items <= 0 || ld < n_items) return fail(h, IFR_ERR_ARG, "ifr_set_item_matrix_dense: null/empty input or ld < n_items");
    if (n_items > (int64_t)1 << 30) return fail(h, IFR_ERR_ARG, "ifr_set_item_matrix_dense: n_items too large");
    IFR_HIP(h, hipSetDevice(h->device));
    const int64_t old_items = h->n_items;
    clear_factor_model(h);
    clear_item_model(h);
    h->n_items = old_items;
    drop_csrs_if_catalogue_changes(h, n_items);
    h->n_items = 0;
    const int chunk = ifr::item_item_chunk(false), pass = ifr::item_item_pass();
    const int64_t ldm = (n_items + pass - 1) / pass * pass;
    IFR_HIP(h, hipMalloc((void **)&h->iiM, sizeof(float) * n_items * ldm));
    if (ldm != n_items) IFR_HIP(h, hipMemset(h->iiM, 0, sizeof(float) * n_items * ldm));
    IFR_HIP(h, hipMemcpy2D(h->iiM, sizeof(float) * ldm, M, sizeof(float) * ld, sizeof(float) * n_items, n_items, hipMemcpyDefault));
    h->ii_ld = ldm;
    h->ii_chunks = (int)((n_items + chunk - 1) / chunk);
    int rc = set_col_scale(h, col_scale, n_items, (int64_t)h->ii_chunks * chunk);
    if (rc) return rc;
    h->n_items = n_items; h->n_users = 0; h->mode = MODE_ITEM_DENSE; h->opt_catalogue = 0;
    return IFR_OK;
}

int ifr_set_item_matrix_dense_f64(ifr_handle h, const double *M, int64_t n_items, int64_t ld, const float *col_scale)
{
    if (!h) return IFR_ERR_ARG;
    if (!M || n_items <= 0 || ld < n_items) return fail(h, IFR_ERR_ARG, "ifr_set_item_matrix_dense_f64: null/empty input or ld < n_items");
    if (n_items > (int64_t)1 << 30) return fail(h, IFR_ERR_ARG, "ifr_set_item_matrix_dense_f64: n_items too large");
    IFR_HIP(h, hipSetDevice(h->device));
    const int64_t old_items = h->n_items;
    clear_factor_model(h);
    clear_item_model(h);
    h->n_items = old_items;
    drop_csrs_if_catalogue_changes(h, n_items);
    h->n_items = 0;
    const int chunk = ifr::item_item_chunk_f64(), pass = ifr::item_item_pass();
    const int64_t ldm = (n_items + pass - 1) / pass * pass;
    IFR_HIP(h, hipMalloc((void **)&h->iiM64, sizeof(double) * n_items * ldm));
    if (ldm != n_items) IFR_HIP(h, hipMemset(h->iiM64, 0, sizeof(double) * n_items * ldm));
    IFR_HIP(h, hipMemcpy2D(h->iiM64, sizeof(double) * ldm, M, sizeof(double) * ld, sizeof(double) * n_items, n_items, hipMemcpyDefault));
    h->ii_ld = ldm;
    h->ii_chunks = (int)((n_items + chunk - 1) / chunk);
    int rc = set_col_scale(h, col_scale, n_items, (int64_t)h->ii_chunks * chunk);
    if (rc) return rc;
    h->n_items = n_items; h->n_users = 0; h->mode = MODE_ITEM_DENSE; h->opt_catalogue = 0;
    return IFR_OK;
}

int ifr_fit_ease(ifr_handle h, const int64_t *x_indptr, const int32_t *x_indices, const float *x_values, int64_t n_users,
                 int64_t n_items, int64_t nnz, double lamb, int posB, double *B_out)
{
    if (!h) return IFR_ERR_ARG;
    if (!x_indptr || n_users < 0 || n_items <= 0 || nnz < 0 || (nnz > 0 && (!x_indices || !x_values)))
        return fail(h, IFR_ERR_ARG, "ifr_fit_ease: null/empty input");
    if (n_items > (int64_t)1 << 30) return fail(h, IFR_ERR_ARG, "ifr_fit_ease: n_items too large");
    if (!(lamb == lamb)) return fail(h, IFR_ERR_ARG, "ifr_fit_ease: lamb is NaN");
    IFR_HIP(h, hipSetDevice(h->device));
    const int64_t old_items = h->n_items;
    clear_factor_model(h);
    clear_item_model(h);
    h->n_items = old_items;
    drop_csrs_if_catalogue_changes(h, n_items);
    h->n_items = 0;
    // X on the device, validated: item indices in range and strictly ascending inside a row, finite values
    int64_t *d_ptr = nullptr;
    int32_t *d_idx = nullptr;
    float *d_val = nullptr;
    int *d_flags = nullptr;
    std::vector<int64_t> hp;
    auto cleanup = [&]() { dfree(d_ptr); dfree(d_idx); dfree(d_val); dfree(d_flags); };
    int rc = copy_csr(h, x_indptr, x_indices, n_users, nnz, &d_ptr, &d_idx, &hp);
    if (rc) { cleanup(); return rc; }
    uint32_t flags[2] = {0, 0};
    bool ok = hipMalloc((void **)&d_val, sizeof(float) * std::max<int64_t>(nnz, 1)) == hipSuccess &&
              (nnz == 0 || hipMemcpy(d_val, x_values, sizeof(float) * nnz, hipMemcpyDefault) == hipSuccess) &&
              hipMalloc((void **)&d_flags, 2 * sizeof(int)) == hipSuccess && hipMemset(d_flags, 0, 2 * sizeof(int)) == hipSuccess &&
              (n_users == 0 || ifr::launch_validate_csr(d_ptr, d_idx, n_users, (int)n_items, 1, d_flags, 0) == 0) &&
              (nnz == 0 || ifr::launch_absmax(d_val, nnz, (uint32_t *)d_flags + 1, 0) == 0) &&
              hipDeviceSynchronize() == hipSuccess && hipMemcpy(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost) == hipSuccess;
    float amax;
    memcpy(&amax, &flags[1], sizeof(float));
    if (!ok || flags[0] != 0 || !std::isfinite(amax)) {
        cleanup();
        return fail(h, ok ? IFR_ERR_ARG : IFR_ERR_HIP,
                    !ok ? "ifr_fit_ease: device set-up failed"
                        : flags[0] ? "ifr_fit_ease: item index outside [0, n_items) or not strictly ascending inside a row"
                                   : "ifr_fit_ease: non-finite rating");
    }
    const bool f64 = h->opt_item_f64 != 0;         // option "item_f64": B stays fp64 and is scored in fp64
    const int chunk = f64 ? ifr::item_item_chunk_f64() : ifr::item_item_chunk(false), pass = ifr::item_item_pass();
    const int64_t ldm = (n_items + pass - 1) / pass * pass;
    const size_t m_bytes = (f64 ? sizeof(double) : sizeof(float)) * (size_t)n_items * (size_t)ldm;
    void *m_dev = nullptr;
    if (hipMalloc(&m_dev, m_bytes) != hipSuccess || hipMemset(m_dev, 0, m_bytes) != hipSuccess) {
        if (m_dev) (void)hipFree(m_dev);
        cleanup(); clear_item_model(h);
        return fail(h, IFR_ERR_HIP, "ifr_fit_ease: device memory for B");
    }
    if (f64) h->iiM64 = (double *)m_dev; else h->iiM = (float *)m_dev;
    char msg[256] = "";
    if (ifr::ease_build(d_ptr, d_idx, d_val, n_users, n_items, lamb, posB, h->iiM, h->iiM64, ldm, B_out, &h->rocblas, msg, sizeof(msg))) {
        cleanup(); clear_item_model(h);
        return fail(h, IFR_ERR_HIP, msg);
    }
    cleanup();
    h->ii_ld = ldm;
    h->ii_chunks = (int)((n_items + chunk - 1) / chunk);
    rc = set_col_scale(h, nullptr, n_items, (int64_t)h->ii_chunks * chunk);
    if (rc) return rc;
    h->n_items = n_items; h->n_users = 0; h->mode = MODE_ITEM_DENSE; h->opt_catalogue = 0;
    return IFR_OK;
}

int ifr_set_item_matrix_csr(ifr_handle h, const int64_t *indptr, const int32_t *indices, const float *values,
                            int64_t n_items, int64_t nnz, const float *col_scale)
{
    if (!h) return IFR_ERR_ARG;
    if (!indptr || n_items <= 0 || nnz < 0 || (nnz > 0 && (!indices || !values)))
        return fail(h, IFR_ERR_ARG, "ifr_set_item_matrix_csr: null/empty input");
    if (n_items > (int64_t)1 << 30) return fail(h, IFR_ERR_ARG, "ifr_set_item_matrix_csr: n_items too large");
    IFR_HIP(h, hipSetDevice(h->device));
    const int64_t old_items = h->n_items;
    clear_factor_model(h);
    clear_item_model(h);
    h->n_items = old_items;
    drop_csrs_if_catalogue_changes(h, n_items);
    h->n_items = 0;
    std::vector<int64_t> hp;
    int64_t *d_ptr = nullptr;
    int rc = copy_csr(h, indptr, indices, n_items, nnz, &d_ptr, &h->ii_idx, &hp);
    if (rc) { dfree(d_ptr); return rc; }
    const int chunk = ifr::item_item_chunk(true);
    const int n_chunks = (int)((n_items + chunk - 1) / chunk);
    int *d_flags = nullptr;                      // [0] = validation flags, [1] = max |value| bits
    auto cleanup = [&]() { dfree(d_ptr); dfree(d_flags); };
    if (hipMalloc((void **)&d_flags, 2 * sizeof(int)) != hipSuccess || hipMemset(d_flags, 0, 2 * sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&h->ii_val, sizeof(float) * (nnz + ifr::item_csr_pad())) != hipSuccess ||
        hipMalloc((void **)&h->ii_cptr, sizeof(int64_t) * n_items * (n_chunks + 1)) != hipSuccess) {
        cleanup(); clear_item_model(h);
        return fail(h, IFR_ERR_NOMEM, "ifr_set_item_matrix_csr: device allocation failed");
    }
    if (nnz > 0 && hipMemcpy(h->ii_val, values, sizeof(float) * nnz, hipMemcpyDefault) != hipSuccess) {
        cleanup(); clear_item_model(h);
        return fail(h, IFR_ERR_HIP, "ifr_set_item_matrix_csr: copy of values failed");
    }
    int lrc = ifr::launch_validate_csr(d_ptr, h->ii_idx, n_items, (int)n_items, 1, d_flags, 0);
    uint32_t flags[2] = {0, 0};
    if (!lrc && (hipDeviceSynchronize() != hipSuccess ||
                 hipMemcpy(flags, d_flags, sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)) lrc = -3;
    if (!lrc && flags[0] == 0) {                 // indices are safe to use from here on
        lrc = ifr::launch_build_chunk_ptr(d_ptr, h->ii_idx, n_items, n_chunks, chunk, h->ii_cptr, 0);
        if (!lrc && col_scale) {                 // fold the column scale into the values (one fp32 rounding each)
            int rc2 = set_col_scale(h, col_scale, n_items, (int64_t)n_chunks * chunk);
            if (rc2) { cleanup(); clear_item_model(h); return rc2; }
            if (nnz > 0) lrc = ifr::launch_scale_csr_values(h->ii_val, h->ii_idx, nnz, h->ii_pen, 0);
        }
        if (!lrc && nnz > 0) lrc = ifr::launch_absmax(h->ii_val, nnz, (uint32_t *)d_flags + 1, 0);
        if (!lrc) {                              // the entries once more as (column, value) pairs, pad included
            const int64_t np = nnz + ifr::item_csr_pad();
            if (hipMalloc((void **)&h->ii_pair, sizeof(uint2) * np) != hipSuccess ||
                hipMemset(h->ii_pair, 0, sizeof(uint2) * np) != hipSuccess) lrc = -4;
            else if (nnz > 0) lrc = ifr::launch_interleave_csr(h->ii_idx, h->ii_val, nnz, h->ii_pair, 0);
        }
        if (!lrc) {                              // entry offsets per 1024-column window, where the table stays modest
            const int n_win = ifr::item_sparse_wave_windows((int)n_items);
            const size_t wbytes = sizeof(uint32_t) * (size_t)n_items * (size_t)ifr::item_sparse_wave_stride(n_win);
            if (wbytes <= ((size_t)1 << 30) && nnz + ifr::item_csr_pad() < ((int64_t)1 << 30) &&
                hipMalloc((void **)&h->ii_cptr_w, wbytes) == hipSuccess) {
                h->ii_n_win = n_win;
                lrc = ifr::launch_build_win_ptr(h->ii_cptr, n_chunks, h->ii_idx, n_items, n_win, h->ii_cptr_w, 0);
                // The wave-per-user kernel reads three entries per row and window (one 16-byte word) and fetches the rest
                // of a longer segment through two dependent loads: it only pays on matrices whose rows spread evenly over
                // the windows (C5's synthetic W; with uniformly drawn columns 16 % of the segments are longer and it runs
                // at half the workgroup kernel's rate).  A wave holds ~25 rows, so with a share f of long segments about
                // 25 f of its windows take that path at ~5 x a window's time: break-even near f = 1 %.  More than 0.3 %
                // long segments: the workgroup kernel serves every user (unless option csr_wave = 2 forces it).
                // A segment may take one, two or three words (3, 6, 9 entries; a lane then keeps 4, 2, 1 history rows): the
                // fewest for which at most 0.3 % of the segments are longer still -- else one word, and the dispatch
                // leaves the matrix to the workgroup kernel.
                unsigned long long *d_long = nullptr, n_long[3] = {0, 0, 0};
                if (!lrc && (hipMalloc((void **)&d_long, sizeof(n_long)) != hipSuccess || hipMemset(d_long, 0, sizeof(n_long)) != hipSuccess)) lrc = -4;
                if (!lrc) lrc = ifr::launch_count_long_segments(h->ii_cptr_w, n_items, n_win, d_long, 0);
                if (!lrc && hipMemcpy(n_long, d_long, sizeof(n_long), hipMemcpyDeviceToHost) != hipSuccess) lrc = -3;
                dfree(d_long);
                const double n_seg = std::max<double>(1.0, (double)n_items * n_win);
                int words = 1;
                while (words < 3 && (double)n_long[words - 1] / n_seg > 0.003) ++words;
                if ((double)n_long[words - 1] / n_seg > 0.003) words = 1;
                // (two / three words cost a lane two / three loads and six / nine LDS adds + exchanges per window whatever the
                // segment holds: only where the segments are reasonably full -- 2 / 4 entries on average -- does that beat
                // the workgroup kernel, whose work follows the entries)
                const double mean_len = (double)nnz / n_seg;
                if ((words == 2 && mean_len < 2.0) || (words == 3 && mean_len < 4.0)) words = 1;
                h->ii_win_words = words;
                h->ii_long_frac = (double)n_long[words - 1] / n_seg;
                // the window words that kernel reads: where they stay within twice the entries' own bytes (or 256 MB) and 4 GB
                const size_t win_bytes = (size_t)16 * (size_t)n_items * (size_t)n_win * (size_t)words;
                const size_t pair_bytes = sizeof(uint2) * (size_t)(nnz + ifr::item_csr_pad());
                if (!lrc && win_bytes <= std::max<size_t>((size_t)256 << 20, 2 * pair_bytes) && win_bytes <= ((size_t)4 << 30) &&
                    hipMalloc(&h->ii_win, win_bytes) == hipSuccess)
                    lrc = ifr::launch_build_win_words(h->ii_cptr_w, h->ii_cptr, n_chunks, (const uint2 *)h->ii_pair, n_items, n_win, words, h->ii_win, 0);
            }
            // (these two tables are optional: an allocation that failed must not linger as the runtime's last error and
            // make the next launch look broken)
            if (h->ii_cptr_w == nullptr || h->ii_win == nullptr) (void)hipGetLastError();
        }
    }
    if (lrc || hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost) != hipSuccess) {
        cleanup(); clear_item_model(h);
        return fail(h, IFR_ERR_HIP, "ifr_set_item_matrix_csr: set-up kernels failed");
    }
    cleanup();
    float maxabs;
    memcpy(&maxabs, &flags[1], sizeof(float));
    if (flags[0] != 0 || !std::isfinite(maxabs)) {
        clear_item_model(h);
        return fail(h, IFR_ERR_ARG, flags[0] & 1 ? "ifr_set_item_matrix_csr: column index outside [0, n_items)"
                                    : flags[0] & 2 ? "ifr_set_item_matrix_csr: columns of a row are not strictly ascending"
                                                   : "ifr_set_item_matrix_csr: non-finite value (after the column scale)");
    }
    h->ii_maxabs = maxabs;
    h->ii_nnz = nnz;
    h->ii_chunks = n_chunks;
    h->n_items = n_items; h->n_users = 0; h->mode = MODE_ITEM_CSR; h->opt_catalogue = 0;
    return IFR_OK;
}

int ifr_set_history_csr(ifr_handle h, const int64_t *indptr, const int32_t *indices, const float *values,
                        int64_t n_rows, int64_t nnz)
{
    if (!h) return IFR_ERR_ARG;
    if (h->mode != MODE_ITEM_DENSE && h->mode != MODE_ITEM_CSR)
        return fail(h, IFR_ERR_STATE, "ifr_set_history_csr: call ifr_set_item_matrix_dense/_csr first");
    IFR_HIP(h, hipSetDevice(h->device));
    dfree(h->hs_ptr); dfree(h->hs_idx); dfree(h->hs_val);
    h->hs_rows = 0; h->hs_nnz = 0; h->hs_l1max = 0.f; h->n_users = 0;
    if (n_rows <= 0 || !indptr) return IFR_OK;
    if (nnz < 0 || (nnz > 0 && (!indices || !values))) return fail(h, IFR_ERR_ARG, "ifr_set_history_csr: null indices/values");
    std::vector<int64_t> hp;
    int rc = copy_csr(h, indptr, indices, n_rows, nnz, &h->hs_ptr, &h->hs_idx, &hp);
    if (rc) return rc;
    int *d_flags = nullptr;
    bool ok = hipMalloc((void **)&h->hs_val, sizeof(float) * std::max<int64_t>(nnz, 1)) == hipSuccess &&
              (nnz == 0 || hipMemcpy(h->hs_val, values, sizeof(float) * nnz, hipMemcpyDefault) == hipSuccess) &&
              hipMalloc((void **)&d_flags, 2 * sizeof(int)) == hipSuccess && hipMemset(d_flags, 0, 2 * sizeof(int)) == hipSuccess;
    uint32_t flags[2] = {0, 0};
    ok = ok && ifr::launch_validate_csr(h->hs_ptr, h->hs_idx, n_rows, (int)h->n_items, 0, d_flags, 0) == 0 &&
         ifr::launch_row_l1_max(h->hs_ptr, h->hs_val, n_rows, (uint32_t *)d_flags + 1, 0) == 0 &&
         hipDeviceSynchronize() == hipSuccess &&
         hipMemcpy(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost) == hipSuccess;
    dfree(d_flags);
    float l1;
    memcpy(&l1, &flags[1], sizeof(float));
    if (!ok || flags[0] != 0 || !std::isfinite(l1)) {
        dfree(h->hs_ptr); dfree(h->hs_idx); dfree(h->hs_val);
        return fail(h, ok ? IFR_ERR_ARG : IFR_ERR_HIP,
                    !ok ? "ifr_set_history_csr: device set-up failed"
                        : flags[0] ? "ifr_set_history_csr: item index outside [0, n_items)" : "ifr_set_history_csr: non-finite weight");
    }
    h->hs_rows = n_rows; h->hs_nnz = nnz; h->hs_l1max = l1; h->n_users = n_rows;
    return IFR_OK;
}

int ifr_fit_rp3beta(ifr_handle h, const int64_t *a_indptr, const int32_t *a_indices, int64_t n_users, int64_t n_items, int64_t nnz,
                    double beta, int64_t *w_nnz)
{
    if (!h) return IFR_ERR_ARG;
    if (!a_indptr || n_users < 0 || n_items <= 0 || nnz < 0 || (nnz > 0 && !a_indices))
        return fail(h, IFR_ERR_ARG, "ifr_fit_rp3beta: null/empty input");
    if (n_items > (int64_t)1 << 30) return fail(h, IFR_ERR_ARG, "ifr_fit_rp3beta: n_items too large");
    if (!(beta == beta)) return fail(h, IFR_ERR_ARG, "ifr_fit_rp3beta: beta is NaN");
    IFR_HIP(h, hipSetDevice(h->device));
    int64_t *d_ptr = nullptr;
    int32_t *d_idx = nullptr;
    int *d_flags = nullptr;
    std::vector<int64_t> hp;
    ifr::Rp3Arrays arr;
    memset(&arr, 0, sizeof(arr));
    auto cleanup = [&]() { dfree(d_ptr); dfree(d_idx); dfree(d_flags); ifr::rp3_free(&arr); };
    int rc = copy_csr(h, a_indptr, a_indices, n_users, nnz, &d_ptr, &d_idx, &hp);
    if (rc) { cleanup(); return rc; }
    uint32_t flag = 0;
    bool ok = hipMalloc((void **)&d_flags, 2 * sizeof(int)) == hipSuccess && hipMemset(d_flags, 0, 2 * sizeof(int)) == hipSuccess &&
              (n_users == 0 || ifr::launch_validate_csr(d_ptr, d_idx, n_users, (int)n_items, 1, d_flags, 0) == 0) &&
              hipDeviceSynchronize() == hipSuccess && hipMemcpy(&flag, d_flags, sizeof(flag), hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok || flag != 0) {
        cleanup();
        return fail(h, ok ? IFR_ERR_ARG : IFR_ERR_HIP,
                    ok ? "ifr_fit_rp3beta: item index outside [0, n_items) or not strictly ascending inside a row"
                       : "ifr_fit_rp3beta: device set-up failed");
    }
    char msg[256] = "";
    if (ifr::rp3_build(d_ptr, d_idx, n_users, n_items, nnz, &h->rocsparse, &arr, msg, sizeof(msg))) {
        cleanup();
        return fail(h, IFR_ERR_HIP, msg);
    }
    // col_scale = 1 / deg(item)^beta from the integer degrees (graph_recommender.py:104-115; degree 0 -> inf)
    std::vector<int64_t> deg(n_items);
    if (hipMemcpy(deg.data(), arr.deg, sizeof(int64_t) * n_items, hipMemcpyDeviceToHost) != hipSuccess) {
        cleanup();
        return fail(h, IFR_ERR_HIP, "ifr_fit_rp3beta: reading the item degrees");
    }
    std::vector<float> pen(n_items);
    for (int64_t i = 0; i < n_items; ++i) pen[i] = (float)(1.0 / std::pow((double)deg[i], beta));
    rc = ifr_set_item_matrix_csr(h, arr.w_ptr, arr.w_idx, arr.w_val, n_items, arr.w_nnz, pen.data());
    if (!rc) rc = ifr_set_history_csr(h, d_ptr, d_idx, arr.p_val32, n_users, nnz);
    if (w_nnz) *w_nnz = arr.w_nnz;
    cleanup();
    return rc;
}

int ifr_item_csr_layout(ifr_handle h, int *win_words, double *long_share)
{
    if (!h) return IFR_ERR_ARG;
    if (h->mode != MODE_ITEM_CSR) return fail(h, IFR_ERR_STATE, "ifr_item_csr_layout: needs a CSR item matrix");
    if (win_words) *win_words = h->ii_win != nullptr ? h->ii_win_words : 0;
    if (long_share) *long_share = h->ii_long_frac;
    return IFR_OK;
}

int ifr_item_fixed_point_shift(ifr_handle h, int *shift)
{
    if (!h || !shift) return IFR_ERR_ARG;
    if (h->mode != MODE_ITEM_CSR || !h->hs_ptr) return fail(h, IFR_ERR_STATE, "ifr_item_fixed_point_shift: needs a CSR item matrix and histories");
    *shift = fixed_point_shift(h);
    return IFR_OK;
}

// item-item launch shared by ifr_rank_topk / ifr_score_dense
static int item_item_params(ifr_handle h, const int32_t *user_ids, int64_t n, ifr::ItemItemParams *p)
{
    if (!h->hs_ptr) return fail(h, IFR_ERR_STATE, "item-item model: histories not set (ifr_set_history_csr)");
    memset(p, 0, sizeof(*p));
    p->M = h->iiM; p->M64 = h->iiM64; p->ldm = h->ii_ld;
    p->cptr = h->ii_cptr; p->m_idx = h->ii_idx; p->m_val = h->ii_val; p->m_pair = h->ii_pair; p->m_nnz = h->ii_nnz;
    p->fx_shift = h->mode == MODE_ITEM_CSR ? fixed_point_shift(h) : 0;
    p->dbg = h->opt_debug;
    p->sp_cap = h->opt_csr_pass > 0 ? h->opt_csr_pass : 0;          // 0: the whole buffer of the kernel that serves K
    p->pen = h->ii_pen;
    p->h_ptr = h->hs_ptr; p->h_idx = h->hs_idx; p->h_val = h->hs_val; p->h_rows = h->hs_rows;
    p->user_ids = user_ids; p->n = n; p->n_items = (int)h->n_items; p->n_chunks = h->ii_chunks;
    p->ex_ptr = h->ex_ptr; p->ex_idx = h->ex_idx; p->ex_rows = h->ex_rows;
    return IFR_OK;
}

static int prof_begin(ifr_handle h, hipStream_t s, std::pair<hipEvent_t, hipEvent_t> *ev)
{
    if (!h->prof) return IFR_OK;
    if (!h->ev_free.empty()) { *ev = h->ev_free.back(); h->ev_free.pop_back(); }
    else { IFR_HIP(h, hipEventCreate(&ev->first)); IFR_HIP(h, hipEventCreate(&ev->second)); }
    IFR_HIP(h, hipEventRecord(ev->first, s));
    return IFR_OK;
}

static int prof_end(ifr_handle h, hipStream_t s, const std::pair<hipEvent_t, hipEvent_t> &ev, double pairs)
{
    if (!h->prof) return IFR_OK;
    IFR_HIP(h, hipEventRecord(ev.second, s));
    h->ev_used.push_back(ev);
    h->prof_pairs += pairs;
    return IFR_OK;
}

// dense / dense_ld: the same launch also writes every score (dense item matrices: fp32 rows, or the float64 sums of the
// fp64 option) -- one scoring pass for the lists and for the row consumers (AUC)
static int rank_topk_item_item(ifr_handle h, const int32_t *user_ids, int64_t n, int K, int KL, int32_t *out_ids,
                               float *out_scores, hipStream_t s, void *dense = nullptr, int64_t dense_ld = 0)
{
    ifr::ItemItemParams p;
    int rc = item_item_params(h, user_ids, n, &p);
    if (rc) return rc;
    if (dense) {
        if (h->iiM64) p.out64 = (double *)dense; else p.out = (float *)dense;
        p.ld = dense_ld;
    }
    if (n > (int64_t)1 << 24) return fail(h, IFR_ERR_ARG, "ifr_rank_topk: more than 2^24 users in one call, split the user list");
    p.K = K; p.KL = KL;
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (h->mode == MODE_ITEM_CSR) {                  // one workgroup per user, final lists written directly
        if (h->ii_cptr_w != nullptr && h->ii_win != nullptr && (h->opt_csr_wave == 2 || (h->opt_csr_wave == 1 && h->ii_long_frac <= 0.003))) {
            rc = grow(h, &h->ws_long, &h->ws_long_bytes, sizeof(int32_t) * ((size_t)n + 1));
            if (rc) return rc;
            p.cptr_w = h->ii_cptr_w; p.n_win = h->ii_n_win; p.m_win = h->ii_win; p.win_words = h->ii_win_words;
            p.long_cnt = (int *)h->ws_long; p.long_list = (int32_t *)h->ws_long + 1;
        }
        rc = prof_begin(h, s, &ev);
        if (rc) return rc;
        if (ifr::launch_item_sparse_topk(p, out_ids, out_scores, s)) return fail(h, IFR_ERR_HIP, "ifr_rank_topk: item-item launch failed");
        return prof_end(h, s, ev, (double)n * (double)h->n_items);
    }
    const size_t entries = (size_t)n * p.n_chunks * KL;
    const bool f64 = h->iiM64 != nullptr;             // fp64 option: the lists carry the fp64 sums and are ranked on them
    rc = grow(h, &h->ws_part, &h->ws_part_bytes, entries * (f64 ? 12 : 8));
    if (rc) return rc;
    if (f64) {
        p.part_scores64 = (double *)h->ws_part;
        p.part_ids = (int32_t *)((char *)h->ws_part + entries * 8);
    } else {
        p.part_scores = (float *)h->ws_part;
        p.part_ids = (int32_t *)((char *)h->ws_part + entries * 4);
    }
    if (f64 && (size_t)p.n_chunks * KL * 12 > 60000) return fail(h, IFR_ERR_ARG, "ifr_rank_topk: catalogue too large for the merge kernel");
    if ((size_t)p.n_chunks * KL * 8 > 60000) return fail(h, IFR_ERR_ARG, "ifr_rank_topk: catalogue too large for the merge kernel");
    rc = prof_begin(h, s, &ev);
    if (rc) return rc;
    if (ifr::launch_item_item(p, false, true, s)) return fail(h, IFR_ERR_HIP, "ifr_rank_topk: item-item launch failed");
    rc = prof_end(h, s, ev, (double)n * (double)h->n_items);
    if (rc) return rc;
    if (f64) {
        if (ifr::launch_merge_topk_f64(p.part_scores64, p.part_ids, n, p.n_chunks, KL, K, out_ids, out_scores, s))
            return fail(h, IFR_ERR_HIP, "ifr_rank_topk: fp64 merge launch failed");
    } else if (KL == 16)
        ifr::launch_merge_topk<16>(p.part_scores, p.part_ids, n, p.n_chunks, p.n_chunks, K, out_ids, out_scores, s);
    else
        ifr::launch_merge_topk<32>(p.part_scores, p.part_ids, n, p.n_chunks, p.n_chunks, K, out_ids, out_scores, s);
    IFR_HIP(h, hipGetLastError());
    return IFR_OK;
}

// fused score + top-K of a dot-product model; dense != nullptr: the kernel also writes every score
// fixed_ws != nullptr: the workspace is that buffer of fixed_bytes (nothing is allocated, no profiling events)
static int rank_topk_factor(ifr_handle h, const int32_t *user_ids, int64_t n, int K, int KL, int32_t *out_ids, float *out_scores,
                            float *dense, int64_t dense_ld, hipStream_t s, void *fixed_ws = nullptr, size_t fixed_bytes = 0);
// the same for 16 < K <= 128 (candidate buffers + exact selection; synchronises the stream once to look for overflows)
static int rank_topk_wide(ifr_handle h, const int32_t *user_ids, int64_t n, int K, int32_t *out_ids, float *out_scores,
                          float *dense, int64_t dense_ld, hipStream_t s);

int ifr_rank_topk(ifr_handle h, const int32_t *user_ids, int64_t n, int K, int32_t *out_ids, float *out_scores, void *stream)
{
    if (!h) return IFR_ERR_ARG;
    if (h->mode == MODE_NONE) return fail(h, IFR_ERR_STATE, "ifr_rank_topk: no model set");
    if (n == 0) return IFR_OK;
    if (!user_ids || !out_ids || !out_scores || n < 0) return fail(h, IFR_ERR_ARG, "ifr_rank_topk: null/negative argument");
    if (K < 1 || K > IFR_MAX_TOPK) return fail(h, IFR_ERR_ARG, "ifr_rank_topk: K must be in [1, 128]");
    IFR_HIP(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    if (h->mode != MODE_FACTOR) {
        if (h->mode == MODE_ITEM_CSR ? K > IFR_MAX_TOPK_ITEM_CSR : K > IFR_MAX_TOPK_ITEM)
            return fail(h, IFR_ERR_ARG, h->mode == MODE_ITEM_CSR ? "ifr_rank_topk: CSR item matrix: K must be in [1, 128] (larger cut-offs: ifr_score_dense + ifr_select_topk)"
                                                                 : "ifr_rank_topk: dense item matrix: K must be in [1, 32] (larger cut-offs: ifr_rank_topk_dense / ifr_score_dense + ifr_select_topk)");
        return rank_topk_item_item(h, user_ids, n, K, K <= 16 ? 16 : 32, out_ids, out_scores, s);
    }
    if (K > std::min(h->opt_pair_max, h->pair_limit) || (K > 16 && !h->opt_pair_topk)) return rank_topk_wide(h, user_ids, n, K, out_ids, out_scores, nullptr, 0, s);
    return rank_topk_factor(h, user_ids, n, K, 16, out_ids, out_scores, nullptr, 0, s);
}

int ifr_rank_topk_dense(ifr_handle h, const int32_t *user_ids, int64_t n, int K, int32_t *out_ids, float *out_scores,
                        float *dense_out, int64_t ld, void *stream)
{
    if (!h) return IFR_ERR_ARG;
    if (h->mode == MODE_ITEM_DENSE && !h->iiM64) {              // dense fp32 item matrix: lists + rows from one pass over M
        if (n == 0) return IFR_OK;
        if (!user_ids || !out_ids || !out_scores || !dense_out || n < 0) return fail(h, IFR_ERR_ARG, "ifr_rank_topk_dense: null/negative argument");
        if (K < 1 || K > IFR_MAX_TOPK_ITEM) return fail(h, IFR_ERR_ARG, "ifr_rank_topk_dense: K must be in [1, 32] for an item-item model");
        if (ld < h->n_items) return fail(h, IFR_ERR_ARG, "ifr_rank_topk_dense: ld < n_items");
        IFR_HIP(h, hipSetDevice(h->device));
        return rank_topk_item_item(h, user_ids, n, K, K <= 16 ? 16 : 32, out_ids, out_scores, (hipStream_t)stream, dense_out, ld);
    }
    if (h->mode != MODE_FACTOR)
        return fail(h, IFR_ERR_STATE, "ifr_rank_topk_dense: needs a dot-product model or a dense fp32 item matrix (fp64: ifr_rank_topk_dense_f64; CSR: ifr_rank_topk + ifr_score_dense_f64)");
    if (n == 0) return IFR_OK;
    if (!user_ids || !out_ids || !out_scores || !dense_out || n < 0) return fail(h, IFR_ERR_ARG, "ifr_rank_topk_dense: null/negative argument");
    if (K < 1 || K > IFR_MAX_TOPK) return fail(h, IFR_ERR_ARG, "ifr_rank_topk_dense: K must be in [1, 128]");
    if (ld < (int64_t)h->n_tiles * ifr::TILE_I || (ld & 3) || ((uintptr_t)dense_out & 15))
        return fail(h, IFR_ERR_ARG, "ifr_rank_topk_dense: ld must be a multiple of 4 and >= n_items rounded up to 32, dense_out 16-byte aligned");
    IFR_HIP(h, hipSetDevice(h->device));
    if (K > std::min(h->opt_pair_max, h->pair_limit) || (K > 16 && !h->opt_pair_topk)) return rank_topk_wide(h, user_ids, n, K, out_ids, out_scores, dense_out, ld, (hipStream_t)stream);
    return rank_topk_factor(h, user_ids, n, K, 16, out_ids, out_scores, dense_out, ld, (hipStream_t)stream);
}

int ifr_rank_topk_dense_f64(ifr_handle h, const int32_t *user_ids, int64_t n, int K, int32_t *out_ids, float *out_scores,
                            double *dense_out, int64_t ld, void *stream)
{
    if (!h) return IFR_ERR_ARG;
    if (h->mode != MODE_ITEM_DENSE || !h->iiM64) return fail(h, IFR_ERR_STATE, "ifr_rank_topk_dense_f64: needs a float64 dense item matrix");
    if (n == 0) return IFR_OK;
    if (!user_ids || !out_ids || !out_scores || !dense_out || n < 0) return fail(h, IFR_ERR_ARG, "ifr_rank_topk_dense_f64: null/negative argument");
    if (K < 1 || K > IFR_MAX_TOPK_ITEM) return fail(h, IFR_ERR_ARG, "ifr_rank_topk_dense_f64: K must be in [1, 32]");
    if (ld < h->n_items) return fail(h, IFR_ERR_ARG, "ifr_rank_topk_dense_f64: ld < n_items");
    IFR_HIP(h, hipSetDevice(h->device));
    return rank_topk_item_item(h, user_ids, n, K, K <= 16 ? 16 : 32, out_ids, out_scores, (hipStream_t)stream, dense_out, ld);
}

static int redo_from_dense_rows(ifr_handle h, const int32_t *user_ids, const int32_t *overflow_slots, uint32_t n_over, int K,
                                int32_t *out_ids, float *out_scores, hipStream_t s);

static int rank_topk_factor(ifr_handle h, const int32_t *user_ids, int64_t n, int K, int KL, int32_t *out_ids, float *out_scores,
                            float *dense, int64_t dense_ld, hipStream_t s, void *fixed_ws, size_t fixed_bytes)
{
    if (!h->U || !h->V || h->n_users <= 0) return fail(h, IFR_ERR_STATE, "ifr_rank_topk: no user factors set");
    const int64_t n_ut = (n + ifr::USERS_PER_BLOCK - 1) / ifr::USERS_PER_BLOCK;
    const size_t lb_bytes = (size_t)n_ut * ifr::USERS_PER_BLOCK * sizeof(uint32_t);
    const size_t chunk_bytes = (size_t)n * 2 * KL * 8;          // two partial lists per user and item chunk
    int64_t cap_chunks = 0;
    if (fixed_ws) {
        if (lb_bytes + chunk_bytes > fixed_bytes) return fail(h, IFR_ERR_ARG, "ifr_rank_topk_noalloc: batch too large for the static workspace (see IFR_NOALLOC_MAX_USERS)");
        cap_chunks = (int64_t)((fixed_bytes - lb_bytes) / chunk_bytes);
    }
    // Pair path (16 < K <= 32, KL = 16): a user's two lists per item chunk hold 32 entries, enough for the top K unless
    // more than 16 of them sit in one list; the merge detects that and those users are ranked again from dense rows.
    // 2 ... 8 item chunks: at least four lists per user (with two, K = 32 would lose an item for 4 users in 10), at
    // most 16 (the all-pairs merge reads 16 entries of each into 256 keys).
    const bool pair = K > KL;
    if (pair && (K > IFR_MAX_TOPK || fixed_ws)) return fail(h, IFR_ERR_ARG, "ifr_rank_topk: internal: list path needs K <= 128 and its own workspace");
    if (pair) cap_chunks = 8;
    const int min_chunks = pair ? std::min(8, std::max(2, (K + h->opt_pair_div - 1) / h->opt_pair_div)) : 1;
    const bool store = dense != nullptr;
    const size_t dyn_lds = (h->opt_debug & 64) ? 30000 : 0;     // dev ablation: extra LDS -> one workgroup per CU
    const topk_fn fn = topk_kernel_kd(h->KD, h->has_base, store, false);
    const GridPlan gp = plan_grid(h, n_ut, resident_slots(h, fn, dyn_lds), min_chunks, cap_chunks);
    const int chunk_tiles = gp.chunk_tiles, n_chunks = gp.n_chunks;
    int n_utA = gp.n_utA, chunk_tilesB = gp.chunk_tilesB, n_chunksB = gp.n_chunksB;
    // users of region A / B and their partial-list entries (2 lists of KL entries per user and item chunk)
    int64_t nA = std::min<int64_t>(n, (int64_t)n_utA * ifr::USERS_PER_BLOCK), nB = n - nA;
    size_t entries = ((size_t)nA * n_chunks + (size_t)nB * n_chunksB) * 2 * KL;
    if (fixed_ws && entries * 8 + lb_bytes > fixed_bytes) {        // the static workspace only holds the one-region plan
        n_utA = (int)n_ut; chunk_tilesB = chunk_tiles; n_chunksB = n_chunks;
        nA = n; nB = 0;
        entries = (size_t)n * n_chunks * 2 * KL;
    }
    const int max_chunks = std::max(n_chunks, n_chunksB);
    if ((size_t)max_chunks * 2 * KL * 8 > 60000)              // checked BEFORE anything is launched
        return fail(h, IFR_ERR_ARG, "ifr_rank_topk: too many item chunks for the merge kernel");
    const int64_t gridA = (int64_t)n_utA * n_chunks, grid = gridA + (n_ut - n_utA) * n_chunksB;
    if (grid > (int64_t)0x7fffffff / ifr::BLOCK) return fail(h, IFR_ERR_ARG, "ifr_rank_topk: batch too large, split the user list");
    void *ws = fixed_ws;
    const size_t over_off = (entries * 8 + lb_bytes + 255) & ~(size_t)255;      // pair path: [count, padding][slots n]
    if (!ws) {
        int rc = grow(h, &h->ws_part, &h->ws_part_bytes, pair ? over_off + 256 + (size_t)n * 4 : entries * 8 + lb_bytes);
        if (rc) return rc;
        ws = h->ws_part;
    }
    uint32_t *over_cnt = pair ? (uint32_t *)((char *)ws + over_off) : nullptr;
    int32_t *over_slots = pair ? (int32_t *)((char *)ws + over_off + 256) : nullptr;
    if (pair) IFR_HIP(h, hipMemsetAsync(over_cnt, 0, 256, s));
    ifr::TopkParams p;
    p.U = h->U; p.V = h->V; p.ibase = h->ibase; p.ubase = h->ubase;
    p.user_ids = user_ids; p.n_users = h->n_users; p.n = n; p.n_items = (int)h->n_items; p.n_tiles = h->n_tiles;
    p.chunk_tiles = chunk_tiles; p.n_chunks = n_chunks; p.K = K;
    p.n_utA = n_utA; p.gridA = (int)gridA; p.chunk_tilesB = chunk_tilesB; p.n_chunksB = n_chunksB;
    p.part_offB = (int64_t)nA * n_chunks * 2 * KL;
    p.ex_ptr = h->ex_ptr; p.ex_cnt = h->ex_cnt; p.ex_pair = h->ex_pair; p.ex_rows = h->ex_rows;
    p.dbg = h->opt_debug;
    p.dense = dense; p.dense_ld = dense_ld;
    p.lbm = nullptr; p.bw = nullptr; p.started = nullptr; p.wide_lists = 0; p.m_rank = 0; p.cand = nullptr; p.cand_cnt = nullptr; p.cand_cap = 0;
    p.stamps = nullptr;
    if ((h->opt_debug & 128) && !fixed_ws) {        // dev-only phase timing, read back by ifr_debug_read_stamps
        int rc2 = grow(h, &h->ws_met, &h->ws_met_bytes, (size_t)grid * 4 * 8 * sizeof(unsigned long long));
        if (rc2) return rc2;
        p.stamps = (unsigned long long *)h->ws_met;
        h->dbg_stamp_count = (size_t)grid * 32;
    }
    p.part_scores = (float *)ws;
    p.part_ids = (int32_t *)((char *)ws + entries * 4);
    // the bound words start at zero on every call (the static workspace keeps the last call's contents)
    p.lb = h->opt_share_bounds ? (uint32_t *)((char *)ws + entries * 8) : nullptr;
    if (p.lb) IFR_HIP(h, hipMemsetAsync(p.lb, 0, lb_bytes, s));

    const bool prof = h->prof && !fixed_ws;
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (prof) {
        if (!h->ev_free.empty()) { ev = h->ev_free.back(); h->ev_free.pop_back(); }
        else { IFR_HIP(h, hipEventCreate(&ev.first)); IFR_HIP(h, hipEventCreate(&ev.second)); }
        IFR_HIP(h, hipEventRecord(ev.first, s));
    }
    {
        void *args[] = {&p};
        IFR_HIP(h, hipLaunchKernel(reinterpret_cast<const void *>(fn), dim3((unsigned)grid), dim3(ifr::BLOCK), args, dyn_lds, s));
    }
    if (prof) {
        IFR_HIP(h, hipEventRecord(ev.second, s));
        h->ev_used.push_back(ev);
        h->prof_pairs += (double)n * (double)h->n_items;
    }
    IFR_HIP(h, hipGetLastError());
    // one merge per region: a user of region A holds 2 * n_chunks lists, one of region B 2 * n_chunksB
    const int64_t offB = p.part_offB;
    if (KL != 16) return fail(h, IFR_ERR_ARG, "ifr_rank_topk: internal: lane lists are 16 entries");
    if (ifr::launch_merge_topk<16>(p.part_scores, p.part_ids, nA, n_chunks * 2, n_chunks * 2, K, out_ids, out_scores, s, over_cnt, over_slots, 0) ||
        ifr::launch_merge_topk<16>(p.part_scores + offB, p.part_ids + offB, nB, n_chunksB * 2, n_chunksB * 2, K, out_ids + nA * K,
                                   out_scores + nA * K, s, over_cnt, over_slots, nA))
        return fail(h, IFR_ERR_ARG, "ifr_rank_topk: internal: merge plan");
    IFR_HIP(h, hipGetLastError());
    if (!pair) return IFR_OK;
    // the one synchronisation of the pair path: how many users must be ranked again
    uint32_t n_over = 0;
    IFR_HIP(h, hipMemcpyAsync(&n_over, over_cnt, sizeof(n_over), hipMemcpyDeviceToHost, s));
    IFR_HIP(h, hipStreamSynchronize(s));
    h->wide_overflows += n_over;
    if (n_over == 0) return IFR_OK;
    if (n >= 64 && (int64_t)n_over * 20 > n) h->pair_limit = std::min(h->pair_limit, K - 1);     // this shape does not suit the path
    return redo_from_dense_rows(h, user_ids, over_slots, n_over, K, out_ids, out_scores, s);
}

int ifr_rank_topk_noalloc(ifr_handle h, const int32_t *user_ids, int64_t n, int K, int32_t *out_ids, float *out_scores, void *stream)
{
    if (!h) return IFR_ERR_ARG;
    if (h->mode != MODE_FACTOR) return fail(h, IFR_ERR_STATE, "ifr_rank_topk_noalloc: needs a dot-product model (ifr_set_factors)");
    if (n == 0) return IFR_OK;
    if (!user_ids || !out_ids || !out_scores || n < 0) return fail(h, IFR_ERR_ARG, "ifr_rank_topk_noalloc: null/negative argument");
    if (K < 1 || K > IFR_NOALLOC_MAX_TOPK) return fail(h, IFR_ERR_ARG, "ifr_rank_topk_noalloc: K must be in [1, 16]");
    if (n > IFR_NOALLOC_MAX_USERS) return fail(h, IFR_ERR_ARG, "ifr_rank_topk_noalloc: more than IFR_NOALLOC_MAX_USERS (65536) users in one call");
    IFR_HIP(h, hipSetDevice(h->device));
    return rank_topk_factor(h, user_ids, n, K, 16, out_ids, out_scores, nullptr, 0, (hipStream_t)stream,
                            h->scratch, (size_t)IFR_NOALLOC_SCRATCH_BYTES);
}


// Users the fused path could not finish (candidate buffer overflow, or a 16-list that may have dropped one of the top K)
// are ranked from dense score rows: exact in every case.  overflow_slots: device, positions in user_ids / the outputs.
static int redo_from_dense_rows(ifr_handle h, const int32_t *user_ids, const int32_t *overflow_slots, uint32_t n_over, int K,
                                int32_t *out_ids, float *out_scores, hipStream_t s)
{
    int rc;
    const int64_t ldr = ((int64_t)h->n_tiles * ifr::TILE_I + 3) / 4 * 4;
    const int64_t rows = std::max<int64_t>(1, std::min<int64_t>(n_over, ((int64_t)1 << 28) / ldr));
    const size_t fb_need = (size_t)rows * ldr * 4 + (size_t)rows * (4 + (size_t)K * 8) + 256;
    rc = grow(h, &h->ws_sort, &h->ws_sort_bytes, fb_need);
    if (rc) return rc;
    float *rows_buf = (float *)h->ws_sort;
    int32_t *fb_users = (int32_t *)((char *)h->ws_sort + (size_t)rows * ldr * 4);
    int32_t *fb_ids = fb_users + rows;
    float *fb_scores = (float *)(fb_ids + rows * K);
    for (int64_t b = 0; b < n_over; b += rows) {
        const int m = (int)std::min<int64_t>(rows, n_over - b);
        hipLaunchKernelGGL(ifr::gather_ids_kernel, dim3((m + 255) / 256), dim3(256), 0, s, user_ids, overflow_slots + b, m, fb_users);
        rc = ifr_score_dense(h, fb_users, m, rows_buf, ldr, s);
        if (rc) return rc;
        if (ifr::launch_select_topk(rows_buf, ldr, fb_users, m, (int)h->n_items, h->ex_ptr, h->ex_idx, h->ex_rows, K, fb_ids, fb_scores, s))
            return fail(h, IFR_ERR_HIP, "ifr_rank_topk: fallback selection failed");
        hipLaunchKernelGGL(ifr::scatter_rows_kernel, dim3((m * K + 255) / 256), dim3(256), 0, s, overflow_slots + b, m, K, fb_ids, fb_scores,
                           out_ids, out_scores);
        IFR_HIP(h, hipGetLastError());
    }
    return IFR_OK;
}

static int rank_topk_wide(ifr_handle h, const int32_t *user_ids, int64_t n, int K, int32_t *out_ids, float *out_scores,
                          float *dense, int64_t dense_ld, hipStream_t s)
{
    if (!h->U || !h->V || h->n_users <= 0) return fail(h, IFR_ERR_STATE, "ifr_rank_topk: no user factors set");
    const int64_t n_ut = (n + ifr::USERS_PER_BLOCK - 1) / ifr::USERS_PER_BLOCK;
    // item chunks: enough lists (2 per chunk) for a bound at list depth <= 12; beyond that as many as fill the chip, but
    // chunks of at least 32 tiles: until its bound words meet (2-3 tiles) a lane appends every score it sees
    const int64_t need_c = (K + 23) / 24;
    const int64_t want = h->opt_chunks > 0 ? h->opt_chunks : (h->opt_target_blocks + n_ut - 1) / n_ut;
    int64_t c = h->opt_chunks > 0 ? std::max(need_c, want) : std::max(need_c, std::min<int64_t>(want, std::min<int64_t>(h->opt_wide_chunk_cap, std::max(1, h->n_tiles / 32))));
    c = std::max<int64_t>(1, std::min<int64_t>(c, std::min<int64_t>(ifr::WIDE_MAX_LISTS / 2, h->n_tiles)));
    const int chunk_tiles = (int)((h->n_tiles + c - 1) / c);
    const int n_chunks = (h->n_tiles + chunk_tiles - 1) / chunk_tiles;
    const int L = 2 * n_chunks;
    // candidate buffer: what a lane appends before the user's bound words meet (about 3 tiles of 16 scores per list) plus
    // 5 x the K (1 + ln(I / K)) entries an ideal running K-th-best threshold lets through (the min-over-lists bound is
    // looser, and late); measured at 26 744 items 1.9 x, at 1 M items 3 x the ideal count
    const double ideal = (double)K * (1.0 + std::log(std::max(1.0, (double)h->n_items / (double)K)));
    const int cap = (int)std::min<double>(ifr::CAND_MAX, std::max(512.0, 48.0 * L + 5.0 * ideal));
    const int64_t grid = n_ut * n_chunks;
    if (grid > (int64_t)0x7fffffff / ifr::BLOCK) return fail(h, IFR_ERR_ARG, "ifr_rank_topk: batch too large, split the user list");
    // workspace: [cand n*cap*8][zeroed: cand_cnt n*4, bw n*4, lbm n*L*4, started n_ut*4, overflow count + statistics][overflow_slots n*4]
    const size_t cand_bytes = (size_t)n * cap * 8;
    const size_t wl_bytes = 0;
    const size_t tail_off = ((size_t)n * 8 + (size_t)n * L * 4 + (size_t)n_ut * 4 + 7) & ~(size_t)7;      // overflow count + statistics, 8-byte aligned
    const size_t zero_bytes = (tail_off + 24 + 255) / 256 * 256;
    const size_t need = cand_bytes + wl_bytes + zero_bytes + (size_t)n * 4;
    int rc = grow(h, &h->ws_part, &h->ws_part_bytes, need);
    if (rc) return rc;
    char *ws = (char *)h->ws_part;
    ifr::TopkParams p;
    memset(&p, 0, sizeof(p));
    p.U = h->U; p.V = h->V; p.ibase = h->ibase; p.ubase = h->ubase;
    p.user_ids = user_ids; p.n_users = h->n_users; p.n = n; p.n_items = (int)h->n_items; p.n_tiles = h->n_tiles;
    p.chunk_tiles = chunk_tiles; p.n_chunks = n_chunks; p.K = K;
    p.n_utA = (int)n_ut; p.gridA = (int)grid; p.chunk_tilesB = chunk_tiles; p.n_chunksB = n_chunks;
    p.ex_ptr = h->ex_ptr; p.ex_cnt = h->ex_cnt; p.ex_pair = h->ex_pair; p.ex_rows = h->ex_rows;
    p.dbg = h->opt_debug & 7;
    p.dense = dense; p.dense_ld = dense_ld;
    p.cand = (uint2 *)ws;
    char *zr = ws + cand_bytes + wl_bytes;
    p.cand_cnt = (uint32_t *)zr;
    p.bw = p.cand_cnt + n;
    p.lbm = p.bw + n;
    p.started = p.lbm + (size_t)n * L;
    uint32_t *overflow_cnt = (uint32_t *)(zr + tail_off);
    unsigned long long *stats = (unsigned long long *)(zr + tail_off + 8);
    int32_t *overflow_slots = (int32_t *)(zr + zero_bytes);
    p.wide_lists = L; p.cand_cap = cap;
    h->dbg_wide_cnt = p.cand_cnt; h->dbg_wide_n = n;
    p.m_rank = (K + L - 1) / L <= 16 ? (K + L - 1) / L : 17;
    IFR_HIP(h, hipMemsetAsync(p.cand_cnt, 0, zero_bytes, s));

    const topk_fn fn = topk_kernel_kd(h->KD, h->has_base, dense != nullptr, true);
    std::pair<hipEvent_t, hipEvent_t> ev;
    rc = prof_begin(h, s, &ev);
    if (rc) return rc;
    {
        void *args[] = {&p};
        IFR_HIP(h, hipLaunchKernel(reinterpret_cast<const void *>(fn), dim3((unsigned)grid), dim3(ifr::BLOCK), args, 0, s));
    }
    rc = prof_end(h, s, ev, (double)n * (double)h->n_items);
    if (rc) return rc;
    ifr::CandParams cp;
    cp.cand = p.cand; cp.cand_cnt = p.cand_cnt; cp.bw = p.bw; cp.cand_cap = cap; cp.n = n; cp.K = K;
    cp.out_ids = out_ids; cp.out_scores = out_scores; cp.overflow_cnt = overflow_cnt; cp.stats = h->opt_wide_stats ? stats : nullptr; cp.overflow_slots = overflow_slots;
    hipLaunchKernelGGL(ifr::select_candidates_kernel, dim3((unsigned)n), dim3(ifr::BLOCK), 0, s, cp);
    IFR_HIP(h, hipGetLastError());
    // users whose buffer overflowed (scores rising with the item index: every item beats the bound of its time) are ranked
    // from dense score rows instead; the one synchronisation of this entry point
    unsigned long long back[3] = {0, 0, 0};            // overflow count (+ padding), candidates appended, candidates kept
    IFR_HIP(h, hipMemcpyAsync(back, overflow_cnt, sizeof(back), hipMemcpyDeviceToHost, s));
    IFR_HIP(h, hipStreamSynchronize(s));
    const uint32_t n_over = (uint32_t)back[0];
    h->wide_overflows += n_over;
    if (h->opt_wide_stats) { h->wide_users += n; h->wide_cand += (double)back[1]; h->wide_kept += (double)back[2]; }
    if (n_over == 0) return IFR_OK;
    return redo_from_dense_rows(h, user_ids, overflow_slots, n_over, K, out_ids, out_scores, s);
}

static int sum_columns(ifr_handle h, const double *per_user, const int32_t *evaluated, int64_t n, int ncol,
                       double *out_sums, hipStream_t s)
{
    if (ncol + 1 > 64) return fail(h, IFR_ERR_ARG, "too many metric columns");
    const int grid = (int)((n + ifr::COLSUM_ROWS - 1) / ifr::COLSUM_ROWS);
    int rc = grow(h, &h->ws_met, &h->ws_met_bytes, sizeof(double) * (size_t)grid * (ncol + 1));
    if (rc) return rc;
    hipLaunchKernelGGL(ifr::colsum_kernel, dim3(grid), dim3(ifr::COLSUM_BLOCK), 0, s, per_user, evaluated, n, ncol, (double *)h->ws_met);
    hipLaunchKernelGGL(ifr::reduce_sums_kernel, dim3(1), dim3(ifr::COLSUM_BLOCK), 0, s, (const double *)h->ws_met, grid, ncol + 1, out_sums);
    IFR_HIP(h, hipGetLastError());
    return IFR_OK;
}

int ifr_eval_topk_metrics(ifr_handle h, const int32_t *user_ids, int64_t n, const int32_t *topk_ids, int K,
                          const int32_t *ks, int n_k, double *out_per_user, int32_t *evaluated, double *out_sums,
                          void *stream)
{
    if (!h) return IFR_ERR_ARG;
    if (!h->tr_ptr) return fail(h, IFR_ERR_STATE, "ifr_eval_topk_metrics: truth CSR not set");
    if (!user_ids || !topk_ids || !ks || !out_per_user || !evaluated || !out_sums || n <= 0)
        return fail(h, IFR_ERR_ARG, "ifr_eval_topk_metrics: null/empty argument");
    if (K < 1 || K > IFR_MAX_TOPK || n_k < 1 || n_k > ifr::MAX_KS)
        return fail(h, IFR_ERR_ARG, "ifr_eval_topk_metrics: K in [1,32], n_k in [1,8]");
    for (int q = 0; q < n_k; ++q)
        if (ks[q] < 1 || ks[q] > K) return fail(h, IFR_ERR_ARG, "ifr_eval_topk_metrics: cut-off outside [1, K]");
    IFR_HIP(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int grid = (int)((n + ifr::BLOCK - 1) / ifr::BLOCK);
    ifr::MetricParams p;
    p.user_ids = user_ids; p.n = n; p.topk = topk_ids; p.K = K; p.n_k = n_k;
    for (int q = 0; q < ifr::MAX_KS; ++q) p.ks[q] = q < n_k ? ks[q] : 0;
    p.tr_ptr = h->tr_ptr; p.tr_idx = h->tr_idx; p.tr_rows = h->tr_rows;
    p.discount = h->discount; p.per_user = out_per_user; p.evaluated = evaluated;
    hipLaunchKernelGGL(ifr::metrics_topk_kernel, dim3(grid), dim3(ifr::BLOCK), 0, s, p);
    IFR_HIP(h, hipGetLastError());
    return sum_columns(h, out_per_user, evaluated, n, IFR_METRICS_PER_K * n_k, out_sums, s);
}

int ifr_eval_ranked_metrics(ifr_handle h, const int32_t *user_ids, int64_t n, const int32_t *ranked, int64_t ld,
                            const int32_t *n_ranked, const int32_t *ks, int n_k, double *out_per_user,
                            int32_t *evaluated, double *out_sums, void *stream)
{
    if (!h) return IFR_ERR_ARG;
    if (!h->tr_ptr) return fail(h, IFR_ERR_STATE, "ifr_eval_ranked_metrics: truth CSR not set");
    if (!user_ids || !ranked || !n_ranked || !ks || !out_per_user || !evaluated || !out_sums || n <= 0 || ld <= 0)
        return fail(h, IFR_ERR_ARG, "ifr_eval_ranked_metrics: null/empty argument");
    if (n_k < 1 || n_k > ifr::MAX_KS) return fail(h, IFR_ERR_ARG, "ifr_eval_ranked_metrics: n_k in [1,8]");
    for (int q = 0; q < n_k; ++q)
        if (ks[q] == 0 || ks[q] < -1) return fail(h, IFR_ERR_ARG, "ifr_eval_ranked_metrics: cut-off must be -1 or > 0");
    IFR_HIP(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    ifr::ListMetricParams p;
    p.user_ids = user_ids; p.n = n; p.ranked = ranked; p.ld = ld; p.n_ranked = n_ranked; p.n_k = n_k;
    for (int q = 0; q < ifr::MAX_KS; ++q) p.ks[q] = q < n_k ? ks[q] : 1;
    p.tr_ptr = h->tr_ptr; p.tr_idx = h->tr_idx; p.tr_rows = h->tr_rows;
    p.per_user = out_per_user; p.evaluated = evaluated;
    hipLaunchKernelGGL(ifr::metrics_list_kernel, dim3((unsigned)n), dim3(ifr::BLOCK), 0, s, p);
    IFR_HIP(h, hipGetLastError());
    return sum_columns(h, out_per_user, evaluated, n, IFR_METRICS_PER_K * n_k + 2, out_sums, s);
}

static int eval_dense_metrics_any(ifr_handle h, const int32_t *user_ids, int64_t n, const void *scores, bool f64, int64_t ld,
                                  const int32_t *ks, int n_k, double *out_per_user, int32_t *evaluated, double *out_sums, void *stream)
{
    if (!h) return IFR_ERR_ARG;
    if (!h->tr_ptr) return fail(h, IFR_ERR_STATE, "ifr_eval_dense_metrics: truth CSR not set");
    const int64_t cat = catalogue_items(h);
    if (!user_ids || !scores || !ks || !out_per_user || !evaluated || !out_sums || n <= 0 || ld < cat)
        return fail(h, IFR_ERR_ARG, "ifr_eval_dense_metrics: null/empty argument or ld < catalogue items");
    if (n_k < 1 || n_k > ifr::MAX_KS) return fail(h, IFR_ERR_ARG, "ifr_eval_dense_metrics: n_k in [1,8]");
    for (int q = 0; q < n_k; ++q)
        if (ks[q] == 0 || ks[q] < -1) return fail(h, IFR_ERR_ARG, "ifr_eval_dense_metrics: cut-off must be -1 or > 0");
    if (h->tr_max_row > (f64 ? ifr::DM_MAX_POS_F64 : ifr::DM_MAX_POS))
        return fail(h, IFR_ERR_ARG, f64 ? "ifr_eval_dense_metrics_f64: more than 2048 positives in one row (use ifr_rank_full_f64 + ifr_eval_ranked_metrics)"
                                        : "ifr_eval_dense_metrics: more than 4096 positives in one row (use ifr_rank_full + ifr_eval_ranked_metrics)");
    if (n > (int64_t)0xffffffffu / ifr::BLOCK) return fail(h, IFR_ERR_ARG, "ifr_eval_dense_metrics: batch too large");
    IFR_HIP(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    ifr::DenseMetricParams p;
    p.scores = scores; p.ld = ld; p.user_ids = user_ids; p.n_items = (int)cat; p.n_k = n_k;
    for (int q = 0; q < ifr::MAX_KS; ++q) p.ks[q] = q < n_k ? ks[q] : 1;
    p.ex_ptr = h->ex_ptr; p.ex_idx = h->ex_idx; p.ex_rows = h->ex_rows;
    p.tr_ptr = h->tr_ptr; p.tr_idx = h->tr_idx; p.tr_rows = h->tr_rows;
    p.per_user = out_per_user; p.evaluated = evaluated;
    int cap = 32;
    while (cap < h->tr_max_row) cap <<= 1;
    p.cap = cap;
    if (f64) hipLaunchKernelGGL(ifr::metrics_dense_kernel<double>, dim3((unsigned)n), dim3(ifr::BLOCK), (size_t)cap * 20 + 16, s, p);
    else hipLaunchKernelGGL(ifr::metrics_dense_kernel<float>, dim3((unsigned)n), dim3(ifr::BLOCK), (size_t)cap * 12 + 16, s, p);
    IFR_HIP(h, hipGetLastError());
    return sum_columns(h, out_per_user, evaluated, n, IFR_METRICS_PER_K * n_k + 2, out_sums, s);
}

int ifr_eval_dense_metrics(ifr_handle h, const int32_t *user_ids, int64_t n, const float *scores, int64_t ld, const int32_t *ks,
                           int n_k, double *out_per_user, int32_t *evaluated, double *out_sums, void *stream)
{
    return eval_dense_metrics_any(h, user_ids, n, scores, false, ld, ks, n_k, out_per_user, evaluated, out_sums, stream);
}

int ifr_eval_dense_metrics_f64(ifr_handle h, const int32_t *user_ids, int64_t n, const double *scores, int64_t ld, const int32_t *ks,
                               int n_k, double *out_per_user, int32_t *evaluated, double *out_sums, void *stream)
{
    return eval_dense_metrics_any(h, user_ids, n, scores, true, ld, ks, n_k, out_per_user, evaluated, out_sums, stream);
}

int ifr_rank_items_f64(ifr_handle h, const double *scores, int64_t n_scores, const int32_t *items, int64_t m, int32_t *out_ids,
                       void *stream)
{
    if (!h) return IFR_ERR_ARG;
    if (m == 0) return IFR_OK;
    if (!scores || !items || !out_ids || m < 0 || n_scores <= 0) return fail(h, IFR_ERR_ARG, "ifr_rank_items_f64: bad argument");
    IFR_HIP(h, hipSetDevice(h->device));
    int rc = ifr::launch_rank_items_f64(scores, n_scores, items, m, out_ids, &h->ws_sort, &h->ws_sort_bytes, (hipStream_t)stream);
    if (rc) return fail(h, rc, "ifr_rank_items_f64: sort failed");
    return IFR_OK;
}

int ifr_auc_dense_f64(ifr_handle h, const int32_t *user_ids, int64_t n, const double *scores, int64_t ld, double *out_auc, void *stream)
{
    if (!h) return IFR_ERR_ARG;
    if (!h->tr_ptr || h->n_items <= 0) return fail(h, IFR_ERR_STATE, "ifr_auc_dense_f64: model / truth CSR not set");
    if (n == 0) return IFR_OK;
    const int64_t cat = catalogue_items(h);
    if (!user_ids || !scores || !out_auc || n < 0 || ld < cat) return fail(h, IFR_ERR_ARG, "ifr_auc_dense_f64: bad argument");
    if (h->tr_max_row > ifr::auc_max_pos_f64()) return fail(h, IFR_ERR_ARG, "ifr_auc_dense_f64: more than 4096 positives in one row");
    IFR_HIP(h, hipSetDevice(h->device));
    if (ifr::launch_auc_f64(scores, ld, user_ids, n, (int)cat, h->ex_ptr, h->ex_idx, h->ex_rows, h->tr_ptr, h->tr_idx, h->tr_rows,
                            h->tr_max_row, out_auc, (hipStream_t)stream))
        return fail(h, IFR_ERR_HIP, "ifr_auc_dense_f64: launch failed");
    return IFR_OK;
}

int ifr_rank_full_f64(ifr_handle h, const int32_t *user_ids, int64_t n, const double *scores, int64_t ld, int32_t *out_ids,
                      int32_t *n_cand, void *stream)
{
    if (!h) return IFR_ERR_ARG;
    if (h->n_items <= 0) return fail(h, IFR_ERR_STATE, "ifr_rank_full_f64: no model set");
    if (n == 0) return IFR_OK;
    const int64_t cat = catalogue_items(h);
    if (!user_ids || !scores || !out_ids || !n_cand || n < 0 || ld < cat) return fail(h, IFR_ERR_ARG, "ifr_rank_full_f64: bad argument");
    if ((uint64_t)n * (uint64_t)cat > 0xffffffffull)
        return fail(h, IFR_ERR_ARG, "ifr_rank_full_f64: more than 2^32 - 1 (user, item) pairs in one call, split the user list");
    IFR_HIP(h, hipSetDevice(h->device));
    int rc = ifr::launch_rank_full_f64(scores, ld, user_ids, n, (int)cat, h->ex_ptr, h->ex_idx, h->ex_rows, out_ids, n_cand,
                                       &h->ws_sort, &h->ws_sort_bytes, (hipStream_t)stream);
    if (rc) return fail(h, rc, "ifr_rank_full_f64: sort failed");
    return IFR_OK;
}

int ifr_select_topk_f64(ifr_handle h, const int32_t *user_ids, int64_t n, const double *scores, int64_t ld, int K, int32_t *out_ids,
                        float *out_scores, void *stream)
{
    if (!h) return IFR_ERR_ARG;
    if (h->n_items <= 0) return fail(h, IFR_ERR_STATE, "ifr_select_topk_f64: no model set");
    if (n == 0) return IFR_OK;
    const int64_t cat = catalogue_items(h);
    if (!user_ids || !scores || !out_ids || !out_scores || n < 0 || ld < cat) return fail(h, IFR_ERR_ARG, "ifr_select_topk_f64: bad argument");
    if (K < 1 || K > ifr::select_topk_max_k()) return fail(h, IFR_ERR_ARG, "ifr_select_topk_f64: K must be in [1, 1024]");
    IFR_HIP(h, hipSetDevice(h->device));
    if (ifr::launch_select_topk_f64(scores, ld, user_ids, n, (int)cat, h->ex_ptr, h->ex_idx, h->ex_rows, K, out_ids, out_scores,
                                    (hipStream_t)stream))
        return fail(h, IFR_ERR_HIP, "ifr_select_topk_f64: launch failed (more than 2^24 rows in one call?)");
    return IFR_OK;
}

int ifr_pad_scores_min_f64(ifr_handle h, double *scores, int64_t n, int64_t ld, void *stream)
{
    if (!h) return IFR_ERR_ARG;
    if (h->n_items <= 0) return fail(h, IFR_ERR_STATE, "ifr_pad_scores_min_f64: no model set");
    const int64_t cat = catalogue_items(h);
    if (n == 0 || cat == h->n_items) return IFR_OK;
    if (!scores || n < 0 || ld < cat) return fail(h, IFR_ERR_ARG, "ifr_pad_scores_min_f64: null argument or ld < catalogue_items");
    IFR_HIP(h, hipSetDevice(h->device));
    if (ifr::launch_pad_rows_min_f64(scores, n, ld, (int)h->n_items, (int)cat, (hipStream_t)stream))
        return fail(h, IFR_ERR_HIP, "ifr_pad_scores_min_f64: launch failed");
    return IFR_OK;
}

int ifr_rank_items(ifr_handle h, const float *scores, int64_t n_scores, const int32_t *items, int64_t m,
                   int32_t *out_ids, void *stream)
{
    if (!h) return IFR_ERR_ARG;
    if (m == 0) return IFR_OK;
    if (!scores || !items || !out_ids || m < 0 || n_scores <= 0) return fail(h, IFR_ERR_ARG, "ifr_rank_items: bad argument");
    IFR_HIP(h, hipSetDevice(h->device));
    int rc = ifr::launch_rank_items(scores, n_scores, items, m, out_ids, &h->ws_sort, &h->ws_sort_bytes, (hipStream_t)stream);
    if (rc) return fail(h, rc, "ifr_rank_items: sort failed");
    return IFR_OK;
}

int ifr_score_dense(ifr_handle h, const int32_t *user_ids, int64_t n, float *out, int64_t ld, void *stream)
{
    if (!h) return IFR_ERR_ARG;
    if (h->mode == MODE_NONE) return fail(h, IFR_ERR_STATE, "ifr_score_dense: no model set");
    if (n == 0) return IFR_OK;
    if (!user_ids || !out || n < 0 || ld < h->n_items) return fail(h, IFR_ERR_ARG, "ifr_score_dense: null argument or ld < n_items");
    IFR_HIP(h, hipSetDevice(h->device));
    if (h->mode != MODE_FACTOR) {
        ifr::ItemItemParams ip;
        int rc = item_item_params(h, user_ids, n, &ip);
        if (rc) return rc;
        if (n > (int64_t)1 << 24) return fail(h, IFR_ERR_ARG, "ifr_score_dense: more than 2^24 users in one call");
        ip.out = out; ip.ld = ld;
        if (ifr::launch_item_item(ip, h->mode == MODE_ITEM_CSR, false, (hipStream_t)stream))
            return fail(h, IFR_ERR_HIP, "ifr_score_dense: item-item launch failed");
        return IFR_OK;
    }
    if (!h->U || !h->V || h->n_users <= 0) return fail(h, IFR_ERR_STATE, "ifr_score_dense: no user factors set");
    int chunk_tiles, n_chunks;
    pick_chunks(h, n, &chunk_tiles, &n_chunks);
    const int64_t n_ut = (n + ifr::USERS_PER_BLOCK - 1) / ifr::USERS_PER_BLOCK;
    const int64_t grid = n_ut * n_chunks;
    if (grid > (int64_t)0xffffffffu / ifr::BLOCK) return fail(h, IFR_ERR_ARG, "ifr_score_dense: batch too large");
    ifr::DenseParams p;
    p.U = h->U; p.V = h->V; p.ibase = h->ibase; p.ubase = h->ubase; p.user_ids = user_ids; p.n_users = h->n_users; p.n = n;
    p.n_items = (int)h->n_items; p.n_tiles = h->n_tiles; p.chunk_tiles = chunk_tiles; p.n_chunks = n_chunks;
    p.out = out; p.ld = ld;
    hipStream_t s = (hipStream_t)stream;
    if (h->KD == 32) launch_dense<32>(h->has_base, p, (int)grid, s);
    else if (h->KD == 64) launch_dense<64>(h->has_base, p, (int)grid, s);
    else if (h->KD == 96) launch_dense<96>(h->has_base, p, (int)grid, s);
    else if (h->KD == 128) launch_dense<128>(h->has_base, p, (int)grid, s);
    else if (h->KD == 160) launch_dense<160>(h->has_base, p, (int)grid, s);
    else if (h->KD == 192) launch_dense<192>(h->has_base, p, (int)grid, s);
    else if (h->KD == 224) launch_dense<224>(h->has_base, p, (int)grid, s);
    else launch_dense<256>(h->has_base, p, (int)grid, s);
    IFR_HIP(h, hipGetLastError());
    return IFR_OK;
}

int ifr_score_dense_f64(ifr_handle h, const int32_t *user_ids, int64_t n, double *out, int64_t ld, void *stream)
{
    if (!h) return IFR_ERR_ARG;
    if (!((h->mode == MODE_ITEM_DENSE && h->iiM64) || h->mode == MODE_ITEM_CSR))
        return fail(h, IFR_ERR_STATE, "ifr_score_dense_f64: needs an fp64 dense item matrix (ifr_set_item_matrix_dense_f64, or ifr_fit_ease with option item_f64) or a CSR item matrix");
    if (n == 0) return IFR_OK;
    if (!user_ids || !out || n < 0 || ld < h->n_items) return fail(h, IFR_ERR_ARG, "ifr_score_dense_f64: null argument or ld < n_items");
    if (n > (int64_t)1 << 24) return fail(h, IFR_ERR_ARG, "ifr_score_dense_f64: more than 2^24 users in one call");
    IFR_HIP(h, hipSetDevice(h->device));
    ifr::ItemItemParams ip;
    int rc = item_item_params(h, user_ids, n, &ip);
    if (rc) return rc;
    ip.out64 = out; ip.ld = ld;
    if (ifr::launch_item_item(ip, h->mode == MODE_ITEM_CSR, false, (hipStream_t)stream)) return fail(h, IFR_ERR_HIP, "ifr_score_dense_f64: launch failed");
    return IFR_OK;
}

int ifr_auc_dense(ifr_handle h, const int32_t *user_ids, int64_t n, const float *scores, int64_t ld, double *out_auc,
                  void *stream)
{
    if (!h) return IFR_ERR_ARG;
    if (!h->tr_ptr || h->n_items <= 0) return fail(h, IFR_ERR_STATE, "ifr_auc_dense: factors / truth CSR not set");
    if (n == 0) return IFR_OK;
    const int64_t cat = catalogue_items(h);
    if (!user_ids || !scores || !out_auc || n < 0 || ld < cat) return fail(h, IFR_ERR_ARG, "ifr_auc_dense: bad argument");
    if (h->tr_max_row > 8192) return fail(h, IFR_ERR_ARG, "ifr_auc_dense: more than 8192 positives in one row");
    IFR_HIP(h, hipSetDevice(h->device));
    int rc = ifr::launch_auc(scores, ld, user_ids, n, (int)cat, h->ex_ptr, h->ex_idx, h->ex_rows, h->tr_ptr,
                             h->tr_idx, h->tr_rows, h->tr_max_row, out_auc, (hipStream_t)stream);
    if (rc) return fail(h, IFR_ERR_HIP, "ifr_auc_dense: launch failed");
    return IFR_OK;
}

int ifr_rank_full(ifr_handle h, const int32_t *user_ids, int64_t n, const float *scores, int64_t ld, int32_t *out_ids,
                  int32_t *n_cand, void *stream)
{
    if (!h) return IFR_ERR_ARG;
    if (h->n_items <= 0) return fail(h, IFR_ERR_STATE, "ifr_rank_full: factors not set");
    if (n == 0) return IFR_OK;
    const int64_t cat = catalogue_items(h);
    if (!user_ids || !scores || !out_ids || !n_cand || n < 0 || ld < cat) return fail(h, IFR_ERR_ARG, "ifr_rank_full: bad argument");
    if ((uint64_t)n * (uint64_t)cat > 0xffffffffull)
        return fail(h, IFR_ERR_ARG, "ifr_rank_full: more than 2^32 - 1 (user, item) pairs in one call, split the user list");
    IFR_HIP(h, hipSetDevice(h->device));
    int rc = ifr::launch_rank_full(scores, ld, user_ids, n, (int)cat, h->ex_ptr, h->ex_idx, h->ex_rows, out_ids,
                                   n_cand, &h->ws_sort, &h->ws_sort_bytes, (hipStream_t)stream);
    if (rc) return fail(h, rc, "ifr_rank_full: sort failed");
    return IFR_OK;
}

int ifr_select_topk(ifr_handle h, const int32_t *user_ids, int64_t n, const float *scores, int64_t ld, int K, int32_t *out_ids,
                    float *out_scores, void *stream)
{
    if (!h) return IFR_ERR_ARG;
    if (h->n_items <= 0) return fail(h, IFR_ERR_STATE, "ifr_select_topk: no model set");
    if (n == 0) return IFR_OK;
    const int64_t cat = catalogue_items(h);
    if (!user_ids || !scores || !out_ids || !out_scores || n < 0 || ld < cat) return fail(h, IFR_ERR_ARG, "ifr_select_topk: bad argument");
    if (K < 1 || K > ifr::select_topk_max_k()) return fail(h, IFR_ERR_ARG, "ifr_select_topk: K must be in [1, 1024]");
    IFR_HIP(h, hipSetDevice(h->device));
    if (ifr::launch_select_topk(scores, ld, user_ids, n, (int)cat, h->ex_ptr, h->ex_idx, h->ex_rows, K, out_ids, out_scores,
                                (hipStream_t)stream))
        return fail(h, IFR_ERR_HIP, "ifr_select_topk: launch failed (more than 2^24 rows in one call?)");
    return IFR_OK;
}

int ifr_pad_scores_min(ifr_handle h, float *scores, int64_t n, int64_t ld, void *stream)
{
    if (!h) return IFR_ERR_ARG;
    if (h->n_items <= 0) return fail(h, IFR_ERR_STATE, "ifr_pad_scores_min: no model set");
    const int64_t cat = catalogue_items(h);
    if (n == 0 || cat == h->n_items) return IFR_OK;
    if (!scores || n < 0 || ld < cat) return fail(h, IFR_ERR_ARG, "ifr_pad_scores_min: null argument or ld < catalogue_items");
    IFR_HIP(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(ifr::pad_rows_min_kernel, dim3((unsigned)n), dim3(ifr::BLOCK), 0, (hipStream_t)stream, scores, ld,
                       (int)h->n_items, (int)cat);
    IFR_HIP(h, hipGetLastError());
    return IFR_OK;
}

int ifr_profile_enable(ifr_handle h, int on)
{
    if (!h) return IFR_ERR_ARG;
    h->prof = on != 0;
    return IFR_OK;
}

int ifr_profile_read(ifr_handle h, double *kernel_ms, int64_t *launches, double *pairs_scored)
{
    if (!h) return IFR_ERR_ARG;
    double ms = 0.0;
    for (auto &e : h->ev_used) {
        IFR_HIP(h, hipEventSynchronize(e.second));
        float t = 0.f;
        IFR_HIP(h, hipEventElapsedTime(&t, e.first, e.second));
        ms += t;
        h->ev_free.push_back(e);
    }
    if (kernel_ms) *kernel_ms = ms;
    if (launches) *launches = (int64_t)h->ev_used.size();
    if (pairs_scored) *pairs_scored = h->prof_pairs;
    h->ev_used.clear();
    h->prof_pairs = 0.0;
    return IFR_OK;
}

int ifr_debug_read_stamps(ifr_handle h, unsigned long long *out, int64_t cap, int64_t *n)
{
    if (!h || !out || !n) return IFR_ERR_ARG;
    IFR_HIP(h, hipDeviceSynchronize());
#ifdef IFR_ABLATE
    if (h->dbg_stamp_count == 0 && cap >= 16 && ifr::item_sparse_read_phases(out) == 0) { *n = 16; return IFR_OK; }
#endif
    *n = (int64_t)std::min<size_t>(h->dbg_stamp_count, (size_t)cap);
    if (*n > 0) IFR_HIP(h, hipMemcpy(out, h->ws_met, sizeof(unsigned long long) * (size_t)*n, hipMemcpyDeviceToHost));
    return IFR_OK;
}

int ifr_wide_overflows(ifr_handle h, int64_t *count)
{
    if (!h || !count) return IFR_ERR_ARG;
    *count = h->wide_overflows;
    return IFR_OK;
}

int ifr_debug_wide_counts(ifr_handle h, uint32_t *out, int64_t cap, int64_t *n)
{
    if (!h || !out || !n) return IFR_ERR_ARG;
    IFR_HIP(h, hipDeviceSynchronize());
    *n = std::min<int64_t>(h->dbg_wide_n, cap);
    if (*n > 0) IFR_HIP(h, hipMemcpy(out, h->dbg_wide_cnt, sizeof(uint32_t) * (size_t)*n, hipMemcpyDeviceToHost));
    return IFR_OK;
}

int ifr_wide_stats(ifr_handle h, int64_t *users, double *candidates, double *kept)
{
    if (!h) return IFR_ERR_ARG;
    if (users) *users = h->wide_users;
    if (candidates) *candidates = h->wide_cand;
    if (kept) *kept = h->wide_kept;
    return IFR_OK;
}

int ifr_factor_chain_order(ifr_handle h, int32_t *perm, int *n)
{
    if (!h || !perm || !n) return IFR_ERR_ARG;
    if (h->mode != MODE_FACTOR || h->KD == 0) return fail(h, IFR_ERR_STATE, "ifr_factor_chain_order: factors not set");
    // MFMA step 4g+e multiplies k = 8g+e (lanes 0-31) then k = 8g+4+e (lanes 32-63).
    int t = 0;
    for (int g = 0; g < h->KD / 8; ++g)
        for (int e = 0; e < 4; ++e) { perm[t++] = 8 * g + e; perm[t++] = 8 * g + 4 + e; }
    *n = h->KD;
    return IFR_OK;
}

}  // extern "C"
