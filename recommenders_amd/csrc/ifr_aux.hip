// ifr_aux.hip -- the two consumers of dense score rows: per-user AUC and the full (k = -1)
// ranking.  Both are HBM-bound passes over [n][n_items] fp32 rows.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace ifr {

constexpr int AUC_BLOCK = 256;
constexpr int AUC_MAX_POS = 8192;

// number of sorted-ascending values strictly greater than s
__device__ __forceinline__ int count_gt(const float *ps, int n, float s)
{
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (ps[mid] > s) hi = mid; else lo = mid + 1; }
    return n - lo;
}

// AUC = #{(p, q): s_p > s_q, p positive, q negative} / (|pos| * |neg|), neg = all - excl - pos.
// With excl and pos disjoint, wins = W(all) - W(excl) - W(pos), W(S) = sum_{i in S} #{p: s_p > s_i},
// so no membership test is needed: three coalesced / gathered passes with a binary search each.
__global__ __launch_bounds__(AUC_BLOCK) void auc_kernel(const float *scores, int64_t ld, const int32_t *user_ids,
                                                        int n_items, const int64_t *ex_ptr, const int32_t *ex_idx,
                                                        int64_t ex_rows, const int64_t *tr_ptr, const int32_t *tr_idx,
                                                        int64_t tr_rows, double *out)
{
    extern __shared__ __attribute__((aligned(16))) float ps[];     // power of two >= the handle's longest truth row (launch_auc)
    __shared__ unsigned long long red[AUC_BLOCK / 64];
    const int64_t i = blockIdx.x;
    const int u = user_ids[i];
    const float *row = scores + i * ld;
    int64_t tb = 0, te = 0, eb = 0, ee = 0;
    __shared__ int s_bad;
    if (u >= 0 && u < tr_rows) { tb = tr_ptr[u]; te = tr_ptr[u + 1]; }
    if (ex_ptr != nullptr && u >= 0 && u < ex_rows) { eb = ex_ptr[u]; ee = ex_ptr[u + 1]; }
    const int n_row = (int)(te - tb);
    if (n_row == 0) {
        if (threadIdx.x == 0) out[i] = __longlong_as_double(0x7ff8000000000000LL);
        return;
    }
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    int np2 = 1;
    while (np2 < n_row) np2 <<= 1;
    // truth entries outside the catalogue are absent, as in metrics_dense_kernel: they sort last as +inf pads and
    // leave n_pos, so they are never read, never win and never count as positives
    {
        int bad = 0;
        for (int q = threadIdx.x; q < np2; q += AUC_BLOCK) {
            float v = INFINITY;
            if (q < n_row) {
                const int it = tr_idx[tb + q];
                if (it >= 0 && it < n_items) v = row[it]; else ++bad;
            }
            ps[q] = v;
        }
        if (bad) atomicAdd(&s_bad, bad);
    }
    __syncthreads();
    const int n_pos = n_row - s_bad;
    if (n_pos == 0) {
        if (threadIdx.x == 0) out[i] = __longlong_as_double(0x7ff8000000000000LL);
        return;
    }
    // bitonic sort ascending in LDS
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int q = threadIdx.x; q < np2; q += AUC_BLOCK) {
                const int x = q ^ j;
                if (x > q) {
                    const float a = ps[q], b = ps[x];
                    const bool up = (q & k) == 0;
                    if ((a > b) == up) { ps[q] = b; ps[x] = a; }
                }
            }
            __syncthreads();
        }
    }
    long long wins = 0;
    // all items: four interleaved branch-free searches per thread (fixed log2(np2) + 1 LDS reads each, so
    // the reads of the four chains overlap) over one 16-byte load of the score row
    int it0 = 0;
    if ((ld & 3) == 0 && (reinterpret_cast<uintptr_t>(scores) & 15) == 0) {
        const int n4 = n_items & ~3;
        // the row is read two iterations ahead: the searches below are chains of dependent LDS reads, and a load issued
        // at the top of its own iteration would add a trip to memory to every one of them
        constexpr int STRIDE = 4 * AUC_BLOCK;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 s_a = 4 * (int)threadIdx.x < n4 ? *reinterpret_cast<const float4 *>(row + 4 * threadIdx.x) : zero4;
        float4 s_b = 4 * (int)threadIdx.x + STRIDE < n4 ? *reinterpret_cast<const float4 *>(row + 4 * threadIdx.x + STRIDE) : zero4;
        for (int it = 4 * threadIdx.x; it < n4; it += STRIDE) {
            const float4 s = s_a;
            s_a = s_b;
            if (it + 2 * STRIDE < n4) s_b = *reinterpret_cast<const float4 *>(row + it + 2 * STRIDE);
            int c0 = 0, c1 = 0, c2 = 0, c3 = 0;                 // elements not greater than s (a prefix of ps)
            for (int step = np2 >> 1; step >= 1; step >>= 1) {
                const float a0 = ps[c0 + step - 1], a1 = ps[c1 + step - 1], a2 = ps[c2 + step - 1], a3 = ps[c3 + step - 1];
                c0 += a0 > s.x ? 0 : step;
                c1 += a1 > s.y ? 0 : step;
                c2 += a2 > s.z ? 0 : step;
                c3 += a3 > s.w ? 0 : step;
            }
            const float a0 = ps[c0], a1 = ps[c1], a2 = ps[c2], a3 = ps[c3];
            c0 += a0 > s.x ? 0 : 1;
            c1 += a1 > s.y ? 0 : 1;
            c2 += a2 > s.z ? 0 : 1;
            c3 += a3 > s.w ? 0 : 1;
            wins += (n_pos - min(c0, n_pos)) + (n_pos - min(c1, n_pos)) + (n_pos - min(c2, n_pos)) + (n_pos - min(c3, n_pos));
        }
        it0 = n4;
    }
    for (int it = it0 + threadIdx.x; it < n_items; it += AUC_BLOCK) wins += count_gt(ps, n_pos, row[it]);
    for (int64_t q = eb + threadIdx.x; q < ee; q += AUC_BLOCK) {
        const int it = ex_idx[q];
        if (it >= 0 && it < n_items) wins -= count_gt(ps, n_pos, row[it]);
    }
    int n_ex = 0;
    for (int64_t q = eb + threadIdx.x; q < ee; q += AUC_BLOCK) {
        const int it = ex_idx[q];
        n_ex += (it >= 0 && it < n_items) ? 1 : 0;
    }
    for (int q = threadIdx.x; q < n_pos; q += AUC_BLOCK) wins -= count_gt(ps, n_pos, ps[q]);
    long long w = wins;                                       // may be transiently negative per thread
    int ne = n_ex;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { w += __shfl_xor(w, off); ne += __shfl_xor(ne, off); }
    __shared__ int red_ne[AUC_BLOCK / 64];
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = (unsigned long long)w; red_ne[threadIdx.x >> 6] = ne; }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long tw = 0; int tne = 0;
        for (int q = 0; q < AUC_BLOCK / 64; ++q) { tw += (long long)red[q]; tne += red_ne[q]; }
        const double n_neg = (double)n_items - (double)tne - (double)n_pos;
        out[i] = (double)tw / ((double)n_pos * n_neg);
    }
}

int launch_auc(const float *scores, int64_t ld, const int32_t *user_ids, int64_t n, int n_items,
               const int64_t *ex_ptr, const int32_t *ex_idx, int64_t ex_rows, const int64_t *tr_ptr,
               const int32_t *tr_idx, int64_t tr_rows, int max_pos, double *out, hipStream_t stream)
{
    if (max_pos > AUC_MAX_POS) return -1;
    int cap = 64;                                   // short truth rows leave LDS for more resident workgroups
    while (cap < max_pos) cap <<= 1;
    hipLaunchKernelGGL(auc_kernel, dim3((unsigned)n), dim3(AUC_BLOCK), sizeof(float) * cap, stream, scores, ld, user_ids, n_items, ex_ptr,
                       ex_idx, ex_rows, tr_ptr, tr_idx, tr_rows, out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ------------------------------------------------------------------------------------
// Full ranking: 64-bit keys (descending-score bits, item id) sorted ascending per user row.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t desc_key(float s)
{
    uint32_t b = __float_as_uint(s);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // ascending-orderable
    return ~b;                                         // descending
}

__global__ void build_keys_kernel(const float *scores, int64_t ld, int n_items, uint64_t *keys)
{
    const int64_t i = blockIdx.y;
    const int it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= n_items) return;
    const float s = scores[i * ld + it];
    keys[i * n_items + it] = ((uint64_t)desc_key(s) << 32) | (uint32_t)it;
}

__global__ void kill_excluded_kernel(const int32_t *user_ids, int n_items, const int64_t *ex_ptr, const int32_t *ex_idx,
                                     int64_t ex_rows, uint64_t *keys, int32_t *n_cand)
{
    const int64_t i = blockIdx.x;
    const int u = user_ids[i];
    int64_t eb = 0, ee = 0;
    if (ex_ptr != nullptr && u >= 0 && u < ex_rows) { eb = ex_ptr[u]; ee = ex_ptr[u + 1]; }
    int c = 0;
    for (int64_t q = eb + threadIdx.x; q < ee; q += blockDim.x) {
        const int it = ex_idx[q];
        if (it >= 0 && it < n_items) { keys[i * n_items + it] = ~0ull; ++c; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
    if (threadIdx.x == 0) n_cand[i] = n_items - c;       // blockDim.x == 64
}

__global__ void keys_to_ids_kernel(const uint64_t *keys, int n_items, int64_t ld, const int32_t *n_cand, int32_t *out)
{
    const int64_t i = blockIdx.y;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ld) return;
    out[i * ld + r] = (r < n_cand[i]) ? (int32_t)(uint32_t)keys[i * n_items + r] : -1;
}

__global__ void build_item_keys_kernel(const float *scores, int64_t n_scores, const int32_t *items, int64_t m, uint64_t *keys)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    const int it = items[q];
    const float s = (it >= 0 && it < n_scores) ? scores[it] : -INFINITY;
    keys[q] = ((uint64_t)desc_key(s) << 32) | (uint32_t)it;
}

__global__ void item_keys_to_ids_kernel(const uint64_t *keys, int64_t m, int32_t *out)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < m) out[q] = (int32_t)(uint32_t)keys[q];
}

// Order an arbitrary candidate list of one user by (score descending, item index ascending).
int launch_rank_items(const float *scores, int64_t n_scores, const int32_t *items, int64_t m, int32_t *out_ids,
                      void **ws, size_t *ws_bytes, hipStream_t stream)
{
    const size_t key_bytes = sizeof(uint64_t) * (size_t)m;
    size_t tmp_bytes = 0;
    uint64_t *null_keys = nullptr;
    if (rocprim::radix_sort_keys(nullptr, tmp_bytes, null_keys, null_keys, (size_t)m, 0, 64, stream) != hipSuccess) return -3;
    const size_t need = 2 * key_bytes + tmp_bytes + 512;
    if (*ws_bytes < need) {
        if (*ws) (void)hipFree(*ws);
        *ws = nullptr; *ws_bytes = 0;
        if (hipMalloc(ws, need) != hipSuccess) return -4;
        *ws_bytes = need;
    }
    uint64_t *k_in = (uint64_t *)*ws;
    uint64_t *k_out = (uint64_t *)((char *)*ws + key_bytes);
    void *tmp = (char *)*ws + 2 * key_bytes;
    const unsigned grid = (unsigned)((m + 255) / 256);
    hipLaunchKernelGGL(build_item_keys_kernel, dim3(grid), dim3(256), 0, stream, scores, n_scores, items, m, k_in);
    if (rocprim::radix_sort_keys(tmp, tmp_bytes, k_in, k_out, (size_t)m, 0, 64, stream) != hipSuccess) return -3;
    hipLaunchKernelGGL(item_keys_to_ids_kernel, dim3(grid), dim3(256), 0, stream, k_out, m, out_ids);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

struct row_offsets {
    int64_t stride;
    __host__ __device__ int64_t operator()(int64_t i) const { return i * stride; }
};

int launch_rank_full(const float *scores, int64_t ld, const int32_t *user_ids, int64_t n, int n_items,
                     const int64_t *ex_ptr, const int32_t *ex_idx, int64_t ex_rows, int32_t *out_ids, int32_t *n_cand,
                     void **ws, size_t *ws_bytes, hipStream_t stream)
{
    if ((size_t)n * (size_t)n_items > 0xffffffffull) return -1;      // rocPRIM's segmented sort takes a 32-bit pair count
    const size_t key_bytes = sizeof(uint64_t) * (size_t)n * n_items;
    auto offs = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int64_t>(0), row_offsets{n_items});
    size_t tmp_bytes = 0;
    uint64_t *null_keys = nullptr;
    if (rocprim::segmented_radix_sort_keys(nullptr, tmp_bytes, null_keys, null_keys, (unsigned)((size_t)n * n_items),
                                           (unsigned)n, offs, offs + 1, 0, 64, stream) != hipSuccess)
        return -3;
    const size_t need = 2 * key_bytes + tmp_bytes + 512;
    if (*ws_bytes < need) {
        if (*ws) (void)hipFree(*ws);
        *ws = nullptr; *ws_bytes = 0;
        if (hipMalloc(ws, need) != hipSuccess) return -4;
        *ws_bytes = need;
    }
    uint64_t *k_in = (uint64_t *)*ws;
    uint64_t *k_out = (uint64_t *)((char *)*ws + key_bytes);
    void *tmp = (char *)*ws + 2 * key_bytes;
    dim3 g((n_items + 255) / 256, (unsigned)n);
    hipLaunchKernelGGL(build_keys_kernel, g, dim3(256), 0, stream, scores, ld, n_items, k_in);
    hipLaunchKernelGGL(kill_excluded_kernel, dim3((unsigned)n), dim3(64), 0, stream, user_ids, n_items, ex_ptr, ex_idx,
                       ex_rows, k_in, n_cand);
    if (rocprim::segmented_radix_sort_keys(tmp, tmp_bytes, k_in, k_out, (unsigned)((size_t)n * n_items), (unsigned)n,
                                           offs, offs + 1, 0, 64, stream) != hipSuccess)
        return -3;
    dim3 g2((unsigned)((ld + 255) / 256), (unsigned)n);
    hipLaunchKernelGGL(keys_to_ids_kernel, g2, dim3(256), 0, stream, k_out, n_items, ld, n_cand, out_ids);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ------------------------------------------------------------------------------------
// Partial ranking of dense score rows: the K best candidates of every row in the order of the full
// ranking above (score descending, item index ascending; excluded items skipped), without sorting
// the row -- Recommender.rank(k != -1) does the same with argpartition + a sort of the head
// (recommender.py:644-649).  One workgroup per row: a three-digit radix select (11 + 11 + 10 bits
// of the descending-orderable score key, LDS histograms, excluded items subtracted afterwards so the
// sweeps need no membership test) finds the key of the K-th candidate; one more sweep in item order
// collects the better ones and, in index order, as many of the equal ones as are still needed;
// the K keys are sorted in LDS.  HBM/L2-bound: four reads of the row.
// ------------------------------------------------------------------------------------
constexpr int SEL_BLOCK = 256;
constexpr int SEL_MAXK = 1024;
constexpr int SEL_BINS = 2048;
constexpr int SEL_UNROLL = 8;

__global__ __launch_bounds__(SEL_BLOCK) void select_topk_kernel(const float *scores, int64_t ld, const int32_t *user_ids,
                                                                int n_items, const int64_t *ex_ptr, const int32_t *ex_idx,
                                                                int64_t ex_rows, int K, int32_t *out_ids, float *out_scores)
{
    __shared__ uint32_t hist[SEL_BINS];
    __shared__ unsigned long long cand[SEL_MAXK];
    __shared__ uint32_t wsum[SEL_BLOCK / 64];
    __shared__ uint32_t s_bin, s_below, s_cnt, s_valid_ex, s_eqn;
    __shared__ int32_t eqbuf[SEL_MAXK];          // items whose key equals the K-th candidate's (collect sweep)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i = blockIdx.x;
    const int u = user_ids[i];
    const float *row = scores + i * ld;
    int64_t eb = 0, ee = 0;
    if (ex_ptr != nullptr && u >= 0 && u < ex_rows) { eb = ex_ptr[u]; ee = ex_ptr[u + 1]; }
    const int64_t n_ex = ee - eb;
    auto excluded = [&](int it) -> bool {
        int64_t lo = 0, hi = n_ex;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (ex_idx[eb + mid] < it) lo = mid + 1; else hi = mid; }
        return lo < n_ex && ex_idx[eb + lo] == it;
    };
    if (tid == 0) { s_cnt = 0; s_valid_ex = 0; s_eqn = 0; }
    __syncthreads();
    {   // candidates of this row: all items minus the valid entries of the exclusion row
        int c = 0;
        for (int64_t q = eb + tid; q < ee; q += SEL_BLOCK) { const int it = ex_idx[q]; c += (it >= 0 && it < n_items) ? 1 : 0; }
        if (c) atomicAdd(&s_valid_ex, (uint32_t)c);
    }
    __syncthreads();
    const int n_cand = n_items - (int)s_valid_ex;
    const int Ke = min(K, n_cand);
    if (Ke <= 0) {
        for (int r = tid; r < K; r += SEL_BLOCK) { out_ids[i * K + r] = -1; out_scores[i * K + r] = -INFINITY; }
        return;
    }
    // ---- the key of the Ke-th candidate, most significant digit first
    uint32_t prefix = 0;                 // digits fixed so far (in place), mask of the fixed bits
    uint32_t pmask = 0;
    uint32_t want = (uint32_t)Ke;        // rank still to find inside the current prefix (1-based)
    const int shifts[3] = {21, 10, 0};
    const int widths[3] = {11, 11, 10};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int sh = shifts[d];
        const uint32_t dmask = (1u << widths[d]) - 1u;
        for (int b = tid; b < SEL_BINS; b += SEL_BLOCK) hist[b] = 0u;
        __syncthreads();
        // SEL_UNROLL loads of a thread are in flight together: with one at a time a 1 M-item row moved at ~1 GB/s per CU
        for (int base = 0; base < n_items; base += SEL_BLOCK * SEL_UNROLL) {
            float v[SEL_UNROLL];
#pragma unroll
            for (int e = 0; e < SEL_UNROLL; ++e) {
                const int it = base + e * SEL_BLOCK + tid;
                v[e] = it < n_items ? row[it] : 0.0f;
            }
#pragma unroll
            for (int e = 0; e < SEL_UNROLL; ++e) {
                const uint32_t k = desc_key(v[e]);
                if (base + e * SEL_BLOCK + tid < n_items && (k & pmask) == prefix) atomicAdd(&hist[(k >> sh) & dmask], 1u);
            }
        }
        for (int64_t q = eb + tid; q < ee; q += SEL_BLOCK) {
            const int it = ex_idx[q];
            if (it >= 0 && it < n_items) {
                const uint32_t k = desc_key(row[it]);
                if ((k & pmask) == prefix) atomicSub(&hist[(k >> sh) & dmask], 1u);
            }
        }
        __syncthreads();
        // first bin whose running count reaches `want`: 8 consecutive bins per thread, block prefix of the sums
        uint32_t loc[SEL_BINS / SEL_BLOCK];
        uint32_t mine = 0;
#pragma unroll
        for (int v = 0; v < SEL_BINS / SEL_BLOCK; ++v) { loc[v] = hist[tid * (SEL_BINS / SEL_BLOCK) + v]; mine += loc[v]; }
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(incl, off); if (lane >= off) incl += t; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t before = incl - mine;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (before < want && want <= before + mine) {          // exactly one thread
            uint32_t run = before;
#pragma unroll
            for (int v = 0; v < SEL_BINS / SEL_BLOCK; ++v) {
                if (run < want && want <= run + loc[v]) { s_bin = (uint32_t)(tid * (SEL_BINS / SEL_BLOCK) + v); s_below = run; }
                run += loc[v];
            }
        }
        __syncthreads();
        prefix |= s_bin << sh;
        pmask |= dmask << sh;
        want -= s_below;
        __syncthreads();
    }
    const uint32_t T = prefix;           // key of the Ke-th candidate; `want` of the candidates with this key are taken
    // ---- collect: better keys anywhere (any order: they are sorted below); of the equal keys the `want` smallest item
    // indices.  One unordered sweep gathers both kinds; almost always exactly `want` items carry the K-th key and all
    // of them are taken.  Ties across the boundary: the smallest `want` of the gathered indices, or -- more tied items
    // than the buffer holds, e.g. a constant row -- one ordered sweep over the row.
    for (int base = 0; base < n_items; base += SEL_BLOCK * SEL_UNROLL) {
        float v[SEL_UNROLL];
#pragma unroll
        for (int e = 0; e < SEL_UNROLL; ++e) {
            const int it = base + e * SEL_BLOCK + tid;
            v[e] = it < n_items ? row[it] : 0.0f;
        }
#pragma unroll
        for (int e = 0; e < SEL_UNROLL; ++e) {
            const int it = base + e * SEL_BLOCK + tid;
            const uint32_t k = desc_key(v[e]);
            // (the index guard of `excluded` only matters for malformed exclusion rows with duplicates, whose double
            // subtraction skews the histograms)
            if (it < n_items && k <= T && !excluded(it)) {
                if (k < T) {
                    const uint32_t sl = atomicAdd(&s_cnt, 1u);
                    if (sl < (uint32_t)SEL_MAXK) cand[sl] = ((unsigned long long)k << 32) | (uint32_t)it;
                } else {
                    const uint32_t sl = atomicAdd(&s_eqn, 1u);
                    if (sl < (uint32_t)SEL_MAXK) eqbuf[sl] = it;
                }
            }
        }
    }
    __syncthreads();
    const uint32_t n_lt = s_cnt, n_eq = s_eqn;
    __syncthreads();
    if (n_eq <= (uint32_t)SEL_MAXK) {
        // entry q is taken if fewer than `want` gathered indices are smaller (all of them when n_eq == want)
        for (uint32_t q = tid; q < n_eq; q += SEL_BLOCK) {
            const int32_t it = eqbuf[q];
            uint32_t smaller = 0;
            if (n_eq > want)
                for (uint32_t r = 0; r < n_eq; ++r) smaller += eqbuf[r] < it ? 1u : 0u;
            if (smaller < want) {
                const uint32_t sl = atomicAdd(&s_cnt, 1u);
                if (sl < (uint32_t)SEL_MAXK) cand[sl] = ((unsigned long long)T << 32) | (uint32_t)it;
            }
        }
    } else {
        uint32_t taken_eq = 0;
        for (int base = 0; base < n_items && taken_eq < want; base += SEL_BLOCK) {
            const int it = base + tid;
            const bool eq = it < n_items && desc_key(row[it]) == T && !excluded(it);
            const unsigned long long bal = __ballot(eq);
            if (lane == 0) wsum[wave] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t before = taken_eq, all = 0;
            for (int w = 0; w < SEL_BLOCK / 64; ++w) { if (w < wave) before += wsum[w]; all += wsum[w]; }
            before += (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (eq && before < want) { const uint32_t sl = atomicAdd(&s_cnt, 1u); if (sl < (uint32_t)SEL_MAXK) cand[sl] = ((unsigned long long)T << 32) | (uint32_t)it; }
            taken_eq += all;
            __syncthreads();
        }
    }
    (void)n_lt;
    __syncthreads();
    // ---- sort the Ke keys ascending (bitonic, padded with the worst key)
    int np2 = 1;
    while (np2 < Ke) np2 <<= 1;
    for (int q = Ke + tid; q < np2; q += SEL_BLOCK) cand[q] = ~0ull;
    __syncthreads();
    for (int kk = 2; kk <= np2; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int q = tid; q < np2; q += SEL_BLOCK) {
                const int x = q ^ j;
                if (x > q) {
                    const unsigned long long a = cand[q], b = cand[x];
                    const bool up = (q & kk) == 0;
                    if ((a > b) == up) { cand[q] = b; cand[x] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int r = tid; r < K; r += SEL_BLOCK) {
        int32_t id = -1;
        float sc = -INFINITY;
        if (r < Ke) {
            const unsigned long long c = cand[r];
            id = (int32_t)(uint32_t)c;
            const uint32_t b = ~(uint32_t)(c >> 32);                       // ascending-orderable bits
            sc = __uint_as_float((b & 0x80000000u) ? (b ^ 0x80000000u) : ~b);
        }
        out_ids[i * K + r] = id;
        out_scores[i * K + r] = sc;
    }
}

int select_topk_max_k() { return SEL_MAXK; }

int launch_select_topk(const float *scores, int64_t ld, const int32_t *user_ids, int64_t n, int n_items, const int64_t *ex_ptr,
                       const int32_t *ex_idx, int64_t ex_rows, int K, int32_t *out_ids, float *out_scores, hipStream_t stream)
{
    if (n <= 0 || n > (int64_t)0xffffffffu / SEL_BLOCK || K < 1 || K > SEL_MAXK) return -1;
    hipLaunchKernelGGL(select_topk_kernel, dim3((unsigned)n), dim3(SEL_BLOCK), 0, stream, scores, ld, user_ids, n_items, ex_ptr,
                       ex_idx, ex_rows, K, out_ids, out_scores);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace ifr
