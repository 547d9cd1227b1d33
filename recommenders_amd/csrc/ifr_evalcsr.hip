// ifr_evalcsr.hip -- the candidate-set builder of ranking_eval on the device.
//
// Reference: cornac/eval_methods/base_method.py:246-283 builds, inside its per-user loop, the positives of the test row
// (rating >= threshold), a dense int mask of everything the user rated >= threshold in train / validation, and from those
// the candidate list.  The batched path needs the same information as two CSRs (SURVEY.md Appendix B rules 1-3):
//     truth[u]     = test items rated >= threshold, clipped to [0, n_items)                       (gt_pos)
//     exclusion[u] = (train U validation items rated >= threshold, < n_items) minus truth[u]      (not ranked)
//     has_pos[u]   = the user has a test item rated >= threshold BEFORE clipping                  (:257-259: evaluated)
// Round 2 built them with SciPy on the host (0.2 s at ML-20M size = 27 x the fused kernel's 7.4 ms, and never run at the
// 600 M-interaction shard).  Here: one wave per user row filters the three raw rows (membership by binary search in the
// raw, column-sorted rows), in two sweeps -- count, prefix sum, fill -- and a rank merge of the train and validation
// runs.  Integer / index work, HBM-bound on the raw rows; results equal the host build bit for bit.
//
// Input rows must be canonical CSR rows (columns strictly ascending inside a row: scipy's sum_duplicates() +
// sort_indices(), what the reference's Dataset.csr_matrix yields).  Ratings are float64 like the reference's
// (data/dataset.py:229-238) or fp32.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <rocprim/device/device_scan.hpp>

#include "ifr_evalcsr.h"

namespace ifr {

constexpr int EC_BLOCK = 256;
constexpr int EC_WAVES = EC_BLOCK / 64;

__device__ __forceinline__ bool ec_rating_ok(const RawCsr &m, int64_t q, double thr)
{
    return (m.f64 ? reinterpret_cast<const double *>(m.val)[q] : (double)reinterpret_cast<const float *>(m.val)[q]) >= thr;
}
__device__ __forceinline__ void ec_row(const RawCsr &m, int64_t u, int64_t *b, int64_t *e)
{
    *b = 0; *e = 0;
    if (m.ptr != nullptr && u >= 0 && u < m.rows) { *b = m.ptr[u]; *e = m.ptr[u + 1]; }
}
// the row holds `item` with a rating >= thr
__device__ __forceinline__ bool ec_row_has_pos(const RawCsr &m, int64_t b, int64_t e, int item, double thr)
{
    int64_t lo = b, hi = e;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (m.idx[mid] < item) lo = mid + 1; else hi = mid; }
    return lo < e && m.idx[lo] == item && ec_rating_ok(m, lo, thr);
}

// One wave per user row.  FILL = false: counts; FILL = true: writes the entries (ranks by ballot prefix, so the order of
// the raw rows -- ascending columns -- is kept).  Exclusion rows are written as two runs: kept train entries, then kept
// validation entries (tmp when a validation matrix exists: merge_runs_kernel interleaves them; else the final array).
template <bool FILL>
__global__ __launch_bounds__(EC_BLOCK) void eval_csr_kernel(EvalCsrParams p)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * EC_WAVES + (threadIdx.x >> 6);
    if (i >= p.n_rows) return;
    const int64_t u = p.row0 + i;
    const unsigned long long below = (1ull << lane) - 1ull;
    int64_t tb, te, ab, ae, vb, ve;
    ec_row(p.test, u, &tb, &te);
    ec_row(p.train, u, &ab, &ae);
    ec_row(p.val, u, &vb, &ve);
    // ---- truth row
    int nt = 0;
    bool any_pos = false;
    const int64_t t_out = FILL ? p.tr_ptr[i] : 0;
    for (int64_t q0 = tb; q0 < te; q0 += 64) {
        const int64_t q = q0 + lane;
        bool ok = false, keep = false;
        int item = 0;
        if (q < te) { item = p.test.idx[q]; ok = ec_rating_ok(p.test, q, p.thr); keep = ok && item >= 0 && item < p.n_items; }
        any_pos |= __any(ok) != 0;
        const unsigned long long bal = __ballot(keep);
        if (FILL && keep) p.tr_idx[t_out + nt + __popcll(bal & below)] = item;
        nt += __popcll(bal);
    }
    // ---- exclusion row: train run
    int na = 0;
    const int64_t e_out = FILL ? p.ex_ptr[i] : 0;
    int32_t *const ex_dst = FILL ? (p.val.ptr != nullptr ? p.ex_tmp : p.ex_idx) : nullptr;
    for (int64_t q0 = ab; q0 < ae; q0 += 64) {
        const int64_t q = q0 + lane;
        bool keep = false;
        int item = 0;
        if (q < ae) {
            item = p.train.idx[q];
            keep = ec_rating_ok(p.train, q, p.thr) && item >= 0 && item < p.n_items && !ec_row_has_pos(p.test, tb, te, item, p.thr);
        }
        const unsigned long long bal = __ballot(keep);
        if (FILL && keep) ex_dst[e_out + na + __popcll(bal & below)] = item;
        na += __popcll(bal);
    }
    // ---- validation run: what train does not already exclude
    int nb = 0;
    for (int64_t q0 = vb; q0 < ve; q0 += 64) {
        const int64_t q = q0 + lane;
        bool keep = false;
        int item = 0;
        if (q < ve) {
            item = p.val.idx[q];
            keep = ec_rating_ok(p.val, q, p.thr) && item >= 0 && item < p.n_items && !ec_row_has_pos(p.test, tb, te, item, p.thr) &&
                   !ec_row_has_pos(p.train, ab, ae, item, p.thr);
        }
        const unsigned long long bal = __ballot(keep);
        if (FILL && keep) ex_dst[e_out + na + nb + __popcll(bal & below)] = item;
        nb += __popcll(bal);
    }
    if (!FILL && lane == 0) {
        p.cnt_truth[i + 1] = nt;
        p.cnt_excl[i + 1] = na + nb;
        if (p.run_a != nullptr) p.run_a[i] = na;
        p.has_pos[i] = any_pos ? 1 : 0;
        if (nt > 0) atomicMax(p.max_truth_row, nt);
    }
}

// exclusion row = merge of its two ascending, disjoint runs: final position = own rank + elements of the other run below
__global__ __launch_bounds__(EC_BLOCK) void merge_runs_kernel(const int64_t *ex_ptr, const int32_t *run_a, const int32_t *tmp,
                                                              int64_t n_rows, int32_t *out)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * EC_WAVES + (threadIdx.x >> 6);
    if (i >= n_rows) return;
    const int64_t b = ex_ptr[i], e = ex_ptr[i + 1];
    const int na = run_a[i], nb = (int)(e - b) - na;
    const int32_t *A = tmp + b, *B = tmp + b + na;
    for (int k = lane; k < na + nb; k += 64) {
        const bool in_a = k < na;
        const int x = in_a ? A[k] : B[k - na];
        const int32_t *other = in_a ? B : A;
        int lo = 0, hi = in_a ? nb : na;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (other[mid] < x) lo = mid + 1; else hi = mid; }
        out[b + (in_a ? k : k - na) + lo] = x;
    }
}

__global__ void ec_zero_first_kernel(int64_t *a, int64_t *b, int *mx)
{
    a[0] = 0; b[0] = 0; *mx = 0;
}

size_t eval_csr_scan_bytes(int64_t n_rows)
{
    size_t bytes = 0;
    int64_t *nul = nullptr;
    (void)rocprim::inclusive_scan(nullptr, bytes, nul, nul, (size_t)(n_rows + 1), rocprim::plus<int64_t>(), (hipStream_t)0);
    return bytes;
}

// count sweep + the two prefix sums: p.cnt_truth / p.cnt_excl (device, [n_rows + 1]) hold the row pointers afterwards
int launch_eval_csr_count(const EvalCsrParams &p, void *scan_tmp, size_t scan_bytes, hipStream_t s)
{
    hipLaunchKernelGGL(ec_zero_first_kernel, dim3(1), dim3(1), 0, s, p.cnt_truth, p.cnt_excl, p.max_truth_row);
    const unsigned grid = (unsigned)((p.n_rows + EC_WAVES - 1) / EC_WAVES);
    if (p.n_rows > 0) hipLaunchKernelGGL(eval_csr_kernel<false>, dim3(grid), dim3(EC_BLOCK), 0, s, p);
    if (rocprim::inclusive_scan(scan_tmp, scan_bytes, p.cnt_truth, p.cnt_truth, (size_t)(p.n_rows + 1), rocprim::plus<int64_t>(), s) != hipSuccess ||
        rocprim::inclusive_scan(scan_tmp, scan_bytes, p.cnt_excl, p.cnt_excl, (size_t)(p.n_rows + 1), rocprim::plus<int64_t>(), s) != hipSuccess)
        return -3;
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_eval_csr_fill(const EvalCsrParams &p, hipStream_t s)
{
    if (p.n_rows <= 0) return 0;
    const unsigned grid = (unsigned)((p.n_rows + EC_WAVES - 1) / EC_WAVES);
    hipLaunchKernelGGL(eval_csr_kernel<true>, dim3(grid), dim3(EC_BLOCK), 0, s, p);
    if (p.val.ptr != nullptr)
        hipLaunchKernelGGL(merge_runs_kernel, dim3(grid), dim3(EC_BLOCK), 0, s, p.ex_ptr, p.run_a, p.ex_tmp, p.n_rows, p.ex_idx);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace ifr
