// ifr_aux.hip -- the consumers of dense score rows: per-user AUC, the full (k = -1) ranking and the partial
// (top-K) ranking.  All are HBM-bound passes over [n][n_items] rows of fp32 scores -- or of fp64 scores, for
// item-item models scored in float64 (the reference ranks the float64 values themselves, recom_ease.py:123-124 ->
// recommender.py:643-652): the fp64 variants order 64-bit keys, so nothing is rounded before it is ranked.
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
constexpr int AUC_MAX_POS_F64 = 4096;          // the sorted positives' scores live in LDS: 32 KB of doubles
constexpr int AUC_BUCKETS = 2048;              // score buckets of the per-user look-up table (8 KB of LDS)
constexpr int AUC_LUT_MIN_POS = 5;             // fewer positives: the binary search is two or three steps anyway

// four consecutive scores of a row, read as 16-byte words
template <typename S> struct Vec4 { S x, y, z, w; };
__device__ __forceinline__ Vec4<float> load4(const float *p)
{
    const float4 v = *reinterpret_cast<const float4 *>(p);
    return {v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ Vec4<double> load4(const double *p)
{
    const double2 a = *reinterpret_cast<const double2 *>(p), b = *reinterpret_cast<const double2 *>(p + 2);
    return {a.x, a.y, b.x, b.y};
}

// number of sorted-ascending values strictly greater than s
template <typename S>
__device__ __forceinline__ int count_gt(const S *ps, int n, S s)
{
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (ps[mid] > s) hi = mid; else lo = mid + 1; }
    return n - lo;
}

// AUC = #{(p, q): s_p > s_q, p positive, q negative} / (|pos| * |neg|), neg = all - excl - pos.
// With excl and pos disjoint, wins = W(all) - W(excl) - W(pos), W(S) = sum_{i in S} #{p: s_p > s_i},
// so no membership test is needed: three coalesced / gathered passes with a binary search each.
template <typename S>
__global__ __launch_bounds__(AUC_BLOCK) void auc_kernel(const S *scores, int64_t ld, const int32_t *user_ids,
                                                        int n_items, const int64_t *ex_ptr, const int32_t *ex_idx,
                                                        int64_t ex_rows, const int64_t *tr_ptr, const int32_t *tr_idx,
                                                        int64_t tr_rows, double *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char auc_lds[];
    S *const ps = reinterpret_cast<S *>(auc_lds);                  // power of two >= the handle's longest truth row (launch_auc)
    __shared__ unsigned long long red[AUC_BLOCK / 64];
    const int64_t i = blockIdx.x;
    const int u = user_ids[i];
    const S *row = scores + i * ld;
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
            S v = (S)INFINITY;
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
                    const S a = ps[q], b = ps[x];
                    const bool up = (q & k) == 0;
                    if ((a > b) == up) { ps[q] = b; ps[x] = a; }
                }
            }
            __syncthreads();
        }
    }
    long long wins = 0;
    int it0 = 0;
    // ---- all items, by a per-user look-up table over score BUCKETS (the common case).  The positives' scores span
    // [lo, hi]; bucket(s) = min(fma(s, scale, off), BUCKETS - 1) converted to an unsigned integer is a monotone
    // non-decreasing function of s (one fma, one min, one conversion: the same instructions for positives and items), so
    //     #{p: s_p > s} = #{p: bucket(s_p) > bucket(s)} + #{p in the bucket of s: s_p > s}
    // exactly.  The table holds the first count and the number of positives per bucket; the second term is a scan of
    // the (almost always empty, rarely one-element) bucket in the sorted array.  A score below lo lands in bucket 0, one
    // above hi in the last one, NaN in the last one (min returns the other operand; every compare is false: no win, as
    // before).  Per score: ~9 VALU and one LDS read instead of a 6 - 7 step binary search of dependent LDS reads.
    __shared__ uint32_t lut[AUC_BUCKETS];                          // low 16 bits: positives in higher buckets; high 16: positives in this one
    __shared__ int scan_tot[AUC_BLOCK / 64];
    const float lo_f = (float)ps[0], hi_f = (float)ps[n_pos - 1];
    const bool use_lut = n_pos >= AUC_LUT_MIN_POS && n_pos < 65536 && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(scores) & 15) == 0 &&
                         lo_f - lo_f == 0.0f && hi_f - hi_f == 0.0f;      // (finite ends; fp32 buckets also for fp64 scores: monotone all the same)
    if (use_lut) {
        float scale = hi_f > lo_f ? (float)(AUC_BUCKETS - 1) / (hi_f - lo_f) : 0.0f;
        if (!(scale - scale == 0.0f)) scale = 0.0f;
        const float off = -lo_f * scale;
        auto bucket = [&](S v) __attribute__((always_inline)) -> uint32_t {
            return (uint32_t)fminf(fmaf((float)v, scale, off), (float)(AUC_BUCKETS - 1));       // (negative -> 0, NaN -> last)
        };
        constexpr int PER = AUC_BUCKETS / AUC_BLOCK;               // consecutive buckets per thread
        for (int q = threadIdx.x; q < AUC_BUCKETS; q += AUC_BLOCK) lut[q] = 0u;
        __syncthreads();
        for (int q = threadIdx.x; q < n_pos; q += AUC_BLOCK) atomicAdd(&lut[bucket(ps[q])], 1u << 16);
        __syncthreads();
        // suffix counts: thread t owns buckets [t PER, (t + 1) PER)
        int mine = 0;
        for (int k = 0; k < PER; ++k) mine += (int)(lut[threadIdx.x * PER + k] >> 16);
        // positives in the buckets of higher threads: a suffix sum over the block (shuffles inside a wave, the four wave
        // totals through LDS)
        int suffix = mine;                                         // inclusive, within the wave
        const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_down(suffix, off); if (ln + off < 64) suffix += o; }
        if (ln == 0) scan_tot[wv] = suffix;
        __syncthreads();
        int above = suffix - mine;
        for (int w2 = wv + 1; w2 < AUC_BLOCK / 64; ++w2) above += scan_tot[w2];
        for (int k = PER - 1; k >= 0; --k) {
            const uint32_t e = lut[threadIdx.x * PER + k];
            lut[threadIdx.x * PER + k] = (e & 0xffff0000u) | (uint32_t)above;
            above += (int)(e >> 16);
        }
        __syncthreads();
        const int n4 = n_items & ~3;
        constexpr int STRIDE = 4 * AUC_BLOCK;
        const Vec4<S> zero4 = {(S)0, (S)0, (S)0, (S)0};
        Vec4<S> s_a = 4 * (int)threadIdx.x < n4 ? load4(row + 4 * threadIdx.x) : zero4;
        Vec4<S> s_b = 4 * (int)threadIdx.x + STRIDE < n4 ? load4(row + 4 * threadIdx.x + STRIDE) : zero4;
        unsigned int w32 = 0;
        // the bucket's positives in the sorted array (almost always none, rarely one)
        auto in_bucket = [&](S v, uint32_t e) __attribute__((always_inline)) {
            const int n_in = (int)(e >> 16), first = n_pos - (int)(e & 0xffffu) - n_in;
            for (int k = 0; k < n_in; ++k) w32 += ps[first + k] > v ? 1u : 0u;
        };
        for (int it = 4 * threadIdx.x; it < n4; it += STRIDE) {
            const Vec4<S> s = s_a;
            s_a = s_b;
            if (it + 2 * STRIDE < n4) s_b = load4(row + it + 2 * STRIDE);
            const uint32_t e0 = lut[bucket(s.x)], e1 = lut[bucket(s.y)], e2 = lut[bucket(s.z)], e3 = lut[bucket(s.w)];      // four reads in flight
            w32 += (e0 & 0xffffu) + (e1 & 0xffffu) + (e2 & 0xffffu) + (e3 & 0xffffu);
            if (__builtin_expect(__any(((e0 | e1 | e2 | e3) >> 16) != 0u) != 0, 0)) {
                in_bucket(s.x, e0); in_bucket(s.y, e1); in_bucket(s.z, e2); in_bucket(s.w, e3);
            }
        }
        wins += (long long)w32;                                    // (< 2^32: at most n_items / 256 * 65535 per thread)
        it0 = n4;
    }
    // all items (few positives / unaligned rows): four interleaved branch-free searches per thread (fixed log2(np2) + 1 LDS reads each, so
    // the reads of the four chains overlap) over one 16-byte load of the score row
    if (!use_lut && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(scores) & 15) == 0) {
        const int n4 = n_items & ~3;
        // the row is read two iterations ahead: the searches below are chains of dependent LDS reads, and a load issued
        // at the top of its own iteration would add a trip to memory to every one of them
        constexpr int STRIDE = 4 * AUC_BLOCK;
        const Vec4<S> zero4 = {(S)0, (S)0, (S)0, (S)0};
        Vec4<S> s_a = 4 * (int)threadIdx.x < n4 ? load4(row + 4 * threadIdx.x) : zero4;
        Vec4<S> s_b = 4 * (int)threadIdx.x + STRIDE < n4 ? load4(row + 4 * threadIdx.x + STRIDE) : zero4;
        for (int it = 4 * threadIdx.x; it < n4; it += STRIDE) {
            const Vec4<S> s = s_a;
            s_a = s_b;
            if (it + 2 * STRIDE < n4) s_b = load4(row + it + 2 * STRIDE);
            int c0 = 0, c1 = 0, c2 = 0, c3 = 0;                 // elements not greater than s (a prefix of ps)
            for (int step = np2 >> 1; step >= 1; step >>= 1) {
                const S a0 = ps[c0 + step - 1], a1 = ps[c1 + step - 1], a2 = ps[c2 + step - 1], a3 = ps[c3 + step - 1];
                c0 += a0 > s.x ? 0 : step;
                c1 += a1 > s.y ? 0 : step;
                c2 += a2 > s.z ? 0 : step;
                c3 += a3 > s.w ? 0 : step;
            }
            const S a0 = ps[c0], a1 = ps[c1], a2 = ps[c2], a3 = ps[c3];
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

template <typename S>
static int launch_auc_t(const S *scores, int64_t ld, const int32_t *user_ids, int64_t n, int n_items,
                        const int64_t *ex_ptr, const int32_t *ex_idx, int64_t ex_rows, const int64_t *tr_ptr,
                        const int32_t *tr_idx, int64_t tr_rows, int max_pos, double *out, hipStream_t stream)
{
    if (max_pos > (sizeof(S) == 8 ? AUC_MAX_POS_F64 : AUC_MAX_POS)) return -1;
    int cap = 64;                                   // short truth rows leave LDS for more resident workgroups
    while (cap < max_pos) cap <<= 1;
    hipLaunchKernelGGL(auc_kernel<S>, dim3((unsigned)n), dim3(AUC_BLOCK), sizeof(S) * cap, stream, scores, ld, user_ids, n_items, ex_ptr,
                       ex_idx, ex_rows, tr_ptr, tr_idx, tr_rows, out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_auc(const float *scores, int64_t ld, const int32_t *user_ids, int64_t n, int n_items,
               const int64_t *ex_ptr, const int32_t *ex_idx, int64_t ex_rows, const int64_t *tr_ptr,
               const int32_t *tr_idx, int64_t tr_rows, int max_pos, double *out, hipStream_t stream)
{
    return launch_auc_t<float>(scores, ld, user_ids, n, n_items, ex_ptr, ex_idx, ex_rows, tr_ptr, tr_idx, tr_rows, max_pos, out, stream);
}
int launch_auc_f64(const double *scores, int64_t ld, const int32_t *user_ids, int64_t n, int n_items,
                   const int64_t *ex_ptr, const int32_t *ex_idx, int64_t ex_rows, const int64_t *tr_ptr,
                   const int32_t *tr_idx, int64_t tr_rows, int max_pos, double *out, hipStream_t stream)
{
    return launch_auc_t<double>(scores, ld, user_ids, n, n_items, ex_ptr, ex_idx, ex_rows, tr_ptr, tr_idx, tr_rows, max_pos, out, stream);
}
int auc_max_pos_f64() { return AUC_MAX_POS_F64; }

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
// The same two rankings for fp64 score rows: the key is the 64-bit descending-orderable image of the double, the item
// index travels as the value of a STABLE radix sort of (key, value) pairs -- items enter in ascending index, so equal
// scores keep that order: score descending, item index ascending, nothing rounded.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t desc_key64(double s)
{
    uint64_t b = (uint64_t)__double_as_longlong(s + 0.0);     // + 0: -0 ranks as +0
    b = (b >> 63) ? ~b : (b | 0x8000000000000000ull);        // ascending-orderable
    return ~b;                                                // descending
}
__device__ __forceinline__ double from_desc_key64(uint64_t k)
{
    const uint64_t b = ~k;
    return __longlong_as_double((long long)((b >> 63) ? (b ^ 0x8000000000000000ull) : ~b));
}

__global__ void build_keys64_kernel(const double *scores, int64_t ld, int n_items, uint64_t *keys, uint32_t *vals)
{
    const int64_t i = blockIdx.y;
    const int it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= n_items) return;
    keys[i * n_items + it] = desc_key64(scores[i * ld + it]);
    vals[i * n_items + it] = (uint32_t)it;
}

__global__ void vals_to_ids_kernel(const uint32_t *vals, int n_items, int64_t ld, const int32_t *n_cand, int32_t *out)
{
    const int64_t i = blockIdx.y;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ld) return;
    out[i * ld + r] = (r < n_cand[i]) ? (int32_t)vals[i * n_items + r] : -1;
}

__global__ void build_item_keys64_kernel(const double *scores, int64_t n_scores, const int32_t *items, int64_t m, uint64_t *keys,
                                         uint64_t *tie)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    const int it = items[q];
    keys[q] = desc_key64((it >= 0 && it < n_scores) ? scores[it] : -INFINITY);
    tie[q] = (uint32_t)it;
}

static int grow_ws(void **ws, size_t *ws_bytes, size_t need)
{
    if (*ws_bytes >= need) return 0;
    if (*ws) (void)hipFree(*ws);
    *ws = nullptr; *ws_bytes = 0;
    if (hipMalloc(ws, need) != hipSuccess) return -4;
    *ws_bytes = need;
    return 0;
}

// The caller's candidate list is in any order: two stable passes -- by item index, then by score key -- give
// (score descending, item index ascending).
int launch_rank_items_f64(const double *scores, int64_t n_scores, const int32_t *items, int64_t m, int32_t *out_ids,
                          void **ws, size_t *ws_bytes, hipStream_t stream)
{
    const size_t kb = sizeof(uint64_t) * (size_t)m;
    size_t tmp_bytes = 0;
    uint64_t *nk = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, nk, nk, nk, nk, (size_t)m, 0, 64, stream) != hipSuccess) return -3;
    if (grow_ws(ws, ws_bytes, 4 * kb + tmp_bytes + 1024)) return -4;
    uint64_t *k_a = (uint64_t *)*ws, *k_b = k_a + m, *t_a = k_b + m, *t_b = t_a + m;
    void *tmp = (char *)*ws + 4 * kb;
    const unsigned grid = (unsigned)((m + 255) / 256);
    hipLaunchKernelGGL(build_item_keys64_kernel, dim3(grid), dim3(256), 0, stream, scores, n_scores, items, m, k_a, t_a);
    // pass 1: by item index (carrying the score key); pass 2: by score key (carrying the item), stable
    if (rocprim::radix_sort_pairs(tmp, tmp_bytes, t_a, t_b, k_a, k_b, (size_t)m, 0, 64, stream) != hipSuccess) return -3;
    if (rocprim::radix_sort_pairs(tmp, tmp_bytes, k_b, k_a, t_b, t_a, (size_t)m, 0, 64, stream) != hipSuccess) return -3;
    hipLaunchKernelGGL(item_keys_to_ids_kernel, dim3(grid), dim3(256), 0, stream, t_a, m, out_ids);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

__global__ void kill_excluded64_kernel(const int32_t *user_ids, int n_items, const int64_t *ex_ptr, const int32_t *ex_idx,
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

int launch_rank_full_f64(const double *scores, int64_t ld, const int32_t *user_ids, int64_t n, int n_items,
                         const int64_t *ex_ptr, const int32_t *ex_idx, int64_t ex_rows, int32_t *out_ids, int32_t *n_cand,
                         void **ws, size_t *ws_bytes, hipStream_t stream)
{
    if ((size_t)n * (size_t)n_items > 0xffffffffull) return -1;
    const size_t N = (size_t)n * n_items;
    auto offs = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int64_t>(0), row_offsets{n_items});
    size_t tmp_bytes = 0;
    uint64_t *nk = nullptr;
    uint32_t *nv = nullptr;
    if (rocprim::segmented_radix_sort_pairs(nullptr, tmp_bytes, nk, nk, nv, nv, (unsigned)N, (unsigned)n, offs, offs + 1, 0, 64,
                                            stream) != hipSuccess)
        return -3;
    if (grow_ws(ws, ws_bytes, 24 * N + tmp_bytes + 1024)) return -4;
    uint64_t *k_in = (uint64_t *)*ws, *k_out = k_in + N;
    uint32_t *v_in = (uint32_t *)(k_out + N), *v_out = v_in + N;
    void *tmp = (char *)*ws + 24 * N;
    dim3 g((n_items + 255) / 256, (unsigned)n);
    hipLaunchKernelGGL(build_keys64_kernel, g, dim3(256), 0, stream, scores, ld, n_items, k_in, v_in);
    hipLaunchKernelGGL(kill_excluded64_kernel, dim3((unsigned)n), dim3(64), 0, stream, user_ids, n_items, ex_ptr, ex_idx, ex_rows,
                       k_in, n_cand);
    if (rocprim::segmented_radix_sort_pairs(tmp, tmp_bytes, k_in, k_out, v_in, v_out, (unsigned)N, (unsigned)n, offs, offs + 1, 0, 64,
                                            stream) != hipSuccess)
        return -3;
    dim3 g2((unsigned)((ld + 255) / 256), (unsigned)n);
    hipLaunchKernelGGL(vals_to_ids_kernel, g2, dim3(256), 0, stream, v_out, n_items, ld, n_cand, out_ids);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// Columns [n_known, n_cat) of every fp64 score row := the row's minimum over the known items (recommender.py:631-635).
__global__ __launch_bounds__(256) void pad_rows_min64_kernel(double *scores, int64_t ld, int n_known, int n_cat)
{
    __shared__ double red[4];
    double *row = scores + (int64_t)blockIdx.x * ld;
    double m = INFINITY;
    for (int c = threadIdx.x; c < n_known; c += 256) m = fmin(m, row[c]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmin(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
    for (int c = n_known + threadIdx.x; c < n_cat; c += 256) row[c] = m;
}
int launch_pad_rows_min_f64(double *scores, int64_t n, int64_t ld, int n_known, int n_cat, hipStream_t stream)
{
    hipLaunchKernelGGL(pad_rows_min64_kernel, dim3((unsigned)n), dim3(256), 0, stream, scores, ld, n_known, n_cat);
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

// descending-orderable key of a score: 32 bits for fp32 rows (three radix digits), 64 bits for fp64 rows (six)
template <typename S> struct SelKey;
template <> struct SelKey<float> {
    typedef uint32_t K;
    static constexpr int NDIG = 3;
    static __device__ __forceinline__ K desc(float s) { return desc_key(s); }
    static __device__ __forceinline__ float to_float(K k)
    {
        const uint32_t b = ~k;                                             // ascending-orderable bits
        return __uint_as_float((b & 0x80000000u) ? (b ^ 0x80000000u) : ~b);
    }
    static __device__ __forceinline__ int shift(int d) { return d == 0 ? 21 : (d == 1 ? 10 : 0); }
    static __device__ __forceinline__ int width(int d) { return d == 2 ? 10 : 11; }
};
template <> struct SelKey<double> {
    typedef uint64_t K;
    static constexpr int NDIG = 6;
    static __device__ __forceinline__ K desc(double s) { return desc_key64(s); }
    static __device__ __forceinline__ float to_float(K k) { return (float)from_desc_key64(k); }
    static __device__ __forceinline__ int shift(int d) { return d == 5 ? 0 : 53 - 11 * d; }     // 53, 42, 31, 20, 9, 0
    static __device__ __forceinline__ int width(int d) { return d == 5 ? 9 : 11; }
};

// fp64 rows: the sweep over the digits stops as soon as the boundary bucket holds exactly the candidates still wanted
// (all of them are taken then) -- after two or three digits on continuous scores, so six digits cost no more sweeps
// than three; true ties run through all six and take the smallest item indices.
template <typename S>
__global__ __launch_bounds__(SEL_BLOCK) void select_topk_kernel(const S *scores, int64_t ld, const int32_t *user_ids,
                                                                int n_items, const int64_t *ex_ptr, const int32_t *ex_idx,
                                                                int64_t ex_rows, int K, int32_t *out_ids, float *out_scores)
{
    typedef typename SelKey<S>::K KT;
    constexpr bool EARLY = sizeof(KT) == 8;
    __shared__ uint32_t hist[SEL_BINS];
    __shared__ KT cand_k[SEL_MAXK];
    __shared__ int32_t cand_i[SEL_MAXK];
    __shared__ uint32_t wsum[SEL_BLOCK / 64];
    __shared__ uint32_t s_bin, s_below, s_bcnt, s_cnt, s_valid_ex, s_eqn;
    __shared__ int32_t eqbuf[SEL_MAXK];          // items whose (masked) key equals the K-th candidate's (collect sweep)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i = blockIdx.x;
    const int u = user_ids[i];
    const S *row = scores + i * ld;
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
    KT prefix = 0;                       // digits fixed so far (in place), mask of the fixed bits
    KT pmask = 0;
    uint32_t want = (uint32_t)Ke;        // rank still to find inside the current prefix (1-based)
    for (int d = 0; d < SelKey<S>::NDIG; ++d) {
        const int sh = SelKey<S>::shift(d);
        const KT dmask = ((KT)1 << SelKey<S>::width(d)) - (KT)1;
        for (int b = tid; b < SEL_BINS; b += SEL_BLOCK) hist[b] = 0u;
        __syncthreads();
        // SEL_UNROLL loads of a thread are in flight together: with one at a time a 1 M-item row moved at ~1 GB/s per CU
        for (int base = 0; base < n_items; base += SEL_BLOCK * SEL_UNROLL) {
            S v[SEL_UNROLL];
#pragma unroll
            for (int e = 0; e < SEL_UNROLL; ++e) {
                const int it = base + e * SEL_BLOCK + tid;
                v[e] = it < n_items ? row[it] : (S)0;
            }
#pragma unroll
            for (int e = 0; e < SEL_UNROLL; ++e) {
                const KT k = SelKey<S>::desc(v[e]);
                if (base + e * SEL_BLOCK + tid < n_items && (k & pmask) == prefix) atomicAdd(&hist[(uint32_t)((k >> sh) & dmask)], 1u);
            }
        }
        for (int64_t q = eb + tid; q < ee; q += SEL_BLOCK) {
            const int it = ex_idx[q];
            if (it >= 0 && it < n_items) {
                const KT k = SelKey<S>::desc(row[it]);
                if ((k & pmask) == prefix) atomicSub(&hist[(uint32_t)((k >> sh) & dmask)], 1u);
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
                if (run < want && want <= run + loc[v]) { s_bin = (uint32_t)(tid * (SEL_BINS / SEL_BLOCK) + v); s_below = run; s_bcnt = loc[v]; }
                run += loc[v];
            }
        }
        __syncthreads();
        prefix |= (KT)s_bin << sh;
        pmask |= dmask << sh;
        want -= s_below;
        const bool whole_bucket = EARLY && s_bcnt == want;      // every candidate of the boundary bucket is taken
        __syncthreads();
        if (whole_bucket) break;
    }
    // (k & pmask) < prefix: better than the boundary bucket; == prefix: in it, `want` of its candidates are taken
    // ---- collect: better keys anywhere (any order: they are sorted below); of the boundary bucket the `want` smallest item
    // indices.  One unordered sweep gathers both kinds; almost always exactly `want` items sit in the bucket and all
    // of them are taken.  Ties across the boundary: the smallest `want` of the gathered indices, or -- more tied items
    // than the buffer holds, e.g. a constant row -- one ordered sweep over the row.
    for (int base = 0; base < n_items; base += SEL_BLOCK * SEL_UNROLL) {
        S v[SEL_UNROLL];
#pragma unroll
        for (int e = 0; e < SEL_UNROLL; ++e) {
            const int it = base + e * SEL_BLOCK + tid;
            v[e] = it < n_items ? row[it] : (S)0;
        }
#pragma unroll
        for (int e = 0; e < SEL_UNROLL; ++e) {
            const int it = base + e * SEL_BLOCK + tid;
            const KT k = SelKey<S>::desc(v[e]);
            const KT km = k & pmask;
            // (the index guard of `excluded` only matters for malformed exclusion rows with duplicates, whose double
            // subtraction skews the histograms)
            if (it < n_items && km <= prefix && !excluded(it)) {
                if (km < prefix) {
                    const uint32_t sl = atomicAdd(&s_cnt, 1u);
                    if (sl < (uint32_t)SEL_MAXK) { cand_k[sl] = k; cand_i[sl] = it; }
                } else {
                    const uint32_t sl = atomicAdd(&s_eqn, 1u);
                    if (sl < (uint32_t)SEL_MAXK) eqbuf[sl] = it;
                }
            }
        }
    }
    __syncthreads();
    const uint32_t n_eq = s_eqn;
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
                if (sl < (uint32_t)SEL_MAXK) { cand_k[sl] = SelKey<S>::desc(row[it]); cand_i[sl] = it; }
            }
        }
    } else {
        uint32_t taken_eq = 0;
        for (int base = 0; base < n_items && taken_eq < want; base += SEL_BLOCK) {
            const int it = base + tid;
            const KT k = it < n_items ? SelKey<S>::desc(row[it]) : (KT)0;
            const bool eq = it < n_items && (k & pmask) == prefix && !excluded(it);
            const unsigned long long bal = __ballot(eq);
            if (lane == 0) wsum[wave] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t before = taken_eq, all = 0;
            for (int w = 0; w < SEL_BLOCK / 64; ++w) { if (w < wave) before += wsum[w]; all += wsum[w]; }
            before += (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (eq && before < want) { const uint32_t sl = atomicAdd(&s_cnt, 1u); if (sl < (uint32_t)SEL_MAXK) { cand_k[sl] = k; cand_i[sl] = it; } }
            taken_eq += all;
            __syncthreads();
        }
    }
    __syncthreads();
    // ---- sort the Ke (key, item) pairs ascending (bitonic, padded with the worst pair)
    int np2 = 1;
    while (np2 < Ke) np2 <<= 1;
    for (int q = Ke + tid; q < np2; q += SEL_BLOCK) { cand_k[q] = ~(KT)0; cand_i[q] = 0x7fffffff; }
    __syncthreads();
    for (int kk = 2; kk <= np2; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int q = tid; q < np2; q += SEL_BLOCK) {
                const int x = q ^ j;
                if (x > q) {
                    const KT ak = cand_k[q], bk = cand_k[x];
                    const int32_t ai = cand_i[q], bi = cand_i[x];
                    const bool up = (q & kk) == 0;
                    const bool gt = ak > bk || (ak == bk && ai > bi);
                    if (gt == up) { cand_k[q] = bk; cand_k[x] = ak; cand_i[q] = bi; cand_i[x] = ai; }
                }
            }
            __syncthreads();
        }
    }
    for (int r = tid; r < K; r += SEL_BLOCK) {
        int32_t id = -1;
        float sc = -INFINITY;
        if (r < Ke) { id = cand_i[r]; sc = SelKey<S>::to_float(cand_k[r]); }
        out_ids[i * K + r] = id;
        out_scores[i * K + r] = sc;
    }
}

int select_topk_max_k() { return SEL_MAXK; }

int launch_select_topk(const float *scores, int64_t ld, const int32_t *user_ids, int64_t n, int n_items, const int64_t *ex_ptr,
                       const int32_t *ex_idx, int64_t ex_rows, int K, int32_t *out_ids, float *out_scores, hipStream_t stream)
{
    if (n <= 0 || n > (int64_t)0xffffffffu / SEL_BLOCK || K < 1 || K > SEL_MAXK) return -1;
    hipLaunchKernelGGL(select_topk_kernel<float>, dim3((unsigned)n), dim3(SEL_BLOCK), 0, stream, scores, ld, user_ids, n_items, ex_ptr,
                       ex_idx, ex_rows, K, out_ids, out_scores);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_select_topk_f64(const double *scores, int64_t ld, const int32_t *user_ids, int64_t n, int n_items, const int64_t *ex_ptr,
                           const int32_t *ex_idx, int64_t ex_rows, int K, int32_t *out_ids, float *out_scores, hipStream_t stream)
{
    if (n <= 0 || n > (int64_t)0xffffffffu / SEL_BLOCK || K < 1 || K > SEL_MAXK) return -1;
    hipLaunchKernelGGL(select_topk_kernel<double>, dim3((unsigned)n), dim3(SEL_BLOCK), 0, stream, scores, ld, user_ids, n_items, ex_ptr,
                       ex_idx, ex_rows, K, out_ids, out_scores);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace ifr
