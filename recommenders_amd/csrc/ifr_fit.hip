// ifr_fit.hip -- item-item model build steps on the device.
//
// EASE closed form, cornac/models/ease/recom_ease.py:75-95:
//     G = X^T X + lamb I          (X = train_set.matrix, CSR [users][items])
//     P = G^-1
//     B = P / (-diag P)  column-wise,  diag B = 0,  posB: negative entries -> 0
//
// RP3-beta item-item form, cornac/models/rp3_beta/graph_recommender.py:24-55, 83-84, 104-115:
//     P_ui = A / deg(u),  P_iu = A^T / deg(i),  W = P_iu . P_ui   (A = binarised train matrix)
// instead of the reference's (U+I)^2 cube P^3: transpose by counting sort (hand-written), the sparse
// product pattern by rocSPARSE SpGEMM, rows sorted by one radix sort of (row, column) keys, values by an ordered
// float64 pass that sums like SciPy's csr_matmat (bit-identical W, run to run and against the host build).
//
// G and P are fp64 like the reference's.  The Gram matrix is a scatter of every user's item pairs
// (nnz_u^2 / 2 fp64 atomics into the stored triangle: HBM-bound, no GEMM to shape: a dense X^T X
// would do users / nnz_u times the flops); the inverse is a plain library factorisation
// (rocSOLVER Cholesky + solve against the identity, 64-bit index API), the final scaling a
// streaming kernel that writes the fp32 matrix the scoring kernels read.
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include <rocsolver/rocsolver.h>
#include <rocsparse/rocsparse.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>

#include "ifr_fit.h"

namespace ifr {

constexpr int FIT_BLOCK = 256;

// one workgroup per user: G[i_a][i_b] += x_a * x_b for the user's item pairs with a >= b (columns
// ascending in a CSR row, so i_a >= i_b: the row-major lower triangle)
__global__ __launch_bounds__(FIT_BLOCK) void gram_pairs_kernel(const int64_t *ptr, const int32_t *idx, const float *val,
                                                               int64_t n_items, double *G)
{
    const int64_t b = ptr[blockIdx.x];
    const int m = (int)(ptr[blockIdx.x + 1] - b);
    const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
    for (int a0 = ty; a0 < m; a0 += FIT_BLOCK / 16) {
        const int64_t row = idx[b + a0];
        const double xa = (double)val[b + a0];
        for (int c0 = tx; c0 <= a0; c0 += 16)
            unsafeAtomicAdd(&G[row * n_items + idx[b + c0]], xa * (double)val[b + c0]);   // exact products, hardware fp64 add
    }
}

__global__ void add_diag_kernel(double *G, int64_t n, double v)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) G[i * n + i] += v;
}

__global__ void identity_kernel(double *S, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) S[i * n + i] = 1.0;
}

__global__ void take_diag_kernel(const double *S, int64_t n, double *d)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = S[i * n + i];
}

// B[j][c] = P[j][c] / (-P[c][c]), zero diagonal, optional clamp; S is overwritten with B (fp64, for the
// caller's copy) and M [n][ldm] receives the fp32 matrix
// (M64 != null: the fp64 matrix [n][ldm] of the fp64 scoring option instead)
__global__ __launch_bounds__(FIT_BLOCK) void ease_scale_kernel(double *S, const double *d, int64_t n, int posB, float *M, double *M64,
                                                               int64_t ldm)
{
    const int64_t j = blockIdx.x;                        // grid.y is limited to 65535: rows go on x
    const int64_t c = (int64_t)blockIdx.y * FIT_BLOCK + threadIdx.x;
    if (c >= n) return;
    double v = S[j * n + c] / (-d[c]);
    if (j == c) v = 0.0;
    if (posB && v < 0.0) v = 0.0;
    S[j * n + c] = v;
    if (M64 != nullptr) M64[j * ldm + c] = v; else M[j * ldm + c] = (float)v;
}

static int fit_fail(char *err, size_t errlen, const char *what, int code)
{
    if (err && errlen) snprintf(err, errlen, "ifr_fit_ease: %s (%d)", what, code);
    return -1;
}

int ease_build(const int64_t *d_ptr, const int32_t *d_idx, const float *d_val, int64_t n_users, int64_t n_items, double lamb,
               int posB, float *M, double *M64, int64_t ldm, double *B_host, void **rocblas_cache, char *err, size_t errlen)
{
    const int64_t n = n_items;
    // HIP: grid.x * block.x < 2^32, grid.y <= 65535
    if (n_users > (int64_t)0xffffffffu / FIT_BLOCK || n > (int64_t)0xffffffffu / FIT_BLOCK)
        return fit_fail(err, errlen, "too many users or items for one launch", 0);
    double *G = nullptr, *S = nullptr, *diag = nullptr;
    int64_t *info = nullptr;
    auto cleanup = [&]() { (void)hipFree(G); (void)hipFree(S); (void)hipFree(diag); (void)hipFree(info); };
    const size_t bytes = sizeof(double) * (size_t)n * (size_t)n;
    hipError_t e;
    if ((e = hipMalloc((void **)&G, bytes)) != hipSuccess || (e = hipMalloc((void **)&S, bytes)) != hipSuccess ||
        (e = hipMalloc((void **)&diag, sizeof(double) * n)) != hipSuccess || (e = hipMalloc((void **)&info, sizeof(int64_t))) != hipSuccess ||
        (e = hipMemset(G, 0, bytes)) != hipSuccess || (e = hipMemset(S, 0, bytes)) != hipSuccess) {
        cleanup();
        return fit_fail(err, errlen, "device memory for two fp64 [items][items] matrices", (int)e);
    }
    if (n_users > 0) hipLaunchKernelGGL(gram_pairs_kernel, dim3((unsigned)n_users), dim3(FIT_BLOCK), 0, 0, d_ptr, d_idx, d_val, n, G);
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(add_diag_kernel, dim3(nb), dim3(256), 0, 0, G, n, lamb);
    hipLaunchKernelGGL(identity_kernel, dim3(nb), dim3(256), 0, 0, S, n);
    if ((e = hipGetLastError()) != hipSuccess) { cleanup(); return fit_fail(err, errlen, "Gram launch", (int)e); }

    rocblas_handle rb = (rocblas_handle)*rocblas_cache;
    if (!rb) {
        if (rocblas_create_handle(&rb) != rocblas_status_success) { cleanup(); return fit_fail(err, errlen, "rocblas_create_handle", 0); }
        *rocblas_cache = rb;
    }
    rocblas_set_stream(rb, 0);
    // the row-major lower triangle is the column-major upper triangle of the same symmetric matrix
    rocblas_status st = rocsolver_dpotrf_64(rb, rocblas_fill_upper, n, G, n, info);
    int64_t h_info = -1;
    if (st == rocblas_status_success) e = hipMemcpy(&h_info, info, sizeof(h_info), hipMemcpyDeviceToHost);
    if (st != rocblas_status_success || e != hipSuccess || h_info != 0) {
        cleanup();
        return fit_fail(err, errlen, h_info > 0 ? "Gram matrix + lamb*I is not positive definite (lamb <= 0?)" : "Cholesky factorisation failed",
                        (int)(h_info > 0 ? h_info : st));
    }
    st = rocsolver_dpotrs_64(rb, rocblas_fill_upper, n, n, G, n, S, n);          // S = G^-1 (symmetric: either storage order)
    if (st != rocblas_status_success) { cleanup(); return fit_fail(err, errlen, "triangular solves failed", (int)st); }
    hipLaunchKernelGGL(take_diag_kernel, dim3(nb), dim3(256), 0, 0, S, n, diag);
    hipLaunchKernelGGL(ease_scale_kernel, dim3((unsigned)n, nb), dim3(FIT_BLOCK), 0, 0, S, diag, n, posB, M, M64, ldm);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess && B_host) e = hipMemcpy(B_host, S, bytes, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fit_fail(err, errlen, "scaling / copy-out", (int)e);
    return 0;
}

void ease_release(void **rocblas_cache)
{
    if (*rocblas_cache) { rocblas_destroy_handle((rocblas_handle)*rocblas_cache); *rocblas_cache = nullptr; }
}

// ------------------------------------------------------------------------------------ RP3-beta
__global__ void column_count_kernel(const int32_t *idx, int64_t nnz, unsigned long long *cnt)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&cnt[idx[e]], 1ull);
}

// one thread per user row: P_ui values (1 / deg(u), fp64 and fp32) and the transposed entries
// (item row <- user column, value 1 / deg(item)); the order inside a transposed row is arbitrary
__global__ void rp3_transpose_kernel(const int64_t *ptr, const int32_t *idx, int64_t n_users, const int64_t *t_ptr,
                                     unsigned long long *cursor, double *p_val, float *p_val32, int32_t *t_idx, double *t_val)
{
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_users) return;
    const int64_t b = ptr[u], e = ptr[u + 1];
    const double pu = 1.0 / (double)(e - b);
    for (int64_t q = b; q < e; ++q) {
        const int32_t i = idx[q];
        p_val[q] = pu;
        p_val32[q] = (float)pu;
        const int64_t pos = t_ptr[i] + (int64_t)atomicAdd(&cursor[i], 1ull);
        t_idx[pos] = (int32_t)u;
        t_val[pos] = 1.0 / (double)(t_ptr[i + 1] - t_ptr[i]);
    }
}

__global__ void rp3_keys_kernel(const int64_t *ptr, const int32_t *idx, int64_t n_rows, uint64_t *keys)
{
    const int64_t r = blockIdx.x;
    for (int64_t q = ptr[r] + threadIdx.x; q < ptr[r + 1]; q += blockDim.x) keys[q] = ((uint64_t)r << 32) | (uint32_t)idx[q];
}

__global__ void rp3_unpack_kernel(const uint64_t *keys, const double *val, int64_t nnz, int32_t *idx, float *val32)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
        idx[e] = (int32_t)(uint32_t)keys[e];
        val32[e] = (float)val[e];
    }
}

// W values in SciPy's summation order.  One wave per item row i: the row's users in ascending order, one after the
// other; for each, the user's items in parallel (distinct columns) add fl(1/deg(i) * 1/deg(u)) at the column's
// position in the row's sorted column list.  A fence between two users keeps their additions to a shared column in
// order, so every W[i, j] is the same left-to-right float64 sum over ascending u that csr_matmat forms
// (graph_recommender.py:83-84 through scipy.sparse) -- bit for bit, on every run.
__global__ __launch_bounds__(64) void rp3_numeric_kernel(const int64_t *t_ptr, const int32_t *t_idx, const int64_t *a_ptr,
                                                         const int32_t *a_idx, const int64_t *c_ptr, const uint64_t *keys,
                                                         double *val, int64_t n_items)
{
    const int lane = threadIdx.x;
    for (int64_t i = blockIdx.x; i < n_items; i += gridDim.x) {
        const int64_t cb = c_ptr[i], ce = c_ptr[i + 1];
        for (int64_t q = cb + lane; q < ce; q += 64) val[q] = 0.0;
        __threadfence();
        const int64_t ub = t_ptr[i], ue = t_ptr[i + 1];
        if (ue == ub) continue;
        const double ai = 1.0 / (double)(ue - ub);
        for (int64_t q = ub; q < ue; ++q) {
            const int u = t_idx[q];
            const int64_t b = a_ptr[u], e = a_ptr[u + 1];
            const double term = ai * (1.0 / (double)(e - b));
            for (int64_t k = b + lane; k < e; k += 64) {
                const uint32_t j = (uint32_t)a_idx[k];
                int64_t lo = cb, hi = ce;                 // column j is in the row: i and j share user u
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if ((uint32_t)keys[mid] < j) lo = mid + 1; else hi = mid;
                }
                if (lo < ce) unsafeAtomicAdd(&val[lo], term);   // performed at L2, in issue order after the fence
            }
            __threadfence();
        }
    }
}

struct to_i64 {
    __host__ __device__ int64_t operator()(unsigned long long x) const { return (int64_t)x; }
};

int rp3_build(const int64_t *a_ptr, const int32_t *a_idx, int64_t n_users, int64_t n_items, int64_t nnz, void **rocsparse_cache,
              Rp3Arrays *out, char *err, size_t errlen)
{
    auto fail = [&](const char *what, int code) {
        if (err && errlen) snprintf(err, errlen, "ifr_fit_rp3beta: %s (%d)", what, code);
        return -1;
    };
    if (n_users > (int64_t)0xffffffffu / 256 || n_items > (int64_t)0xffffffffu / 256 || nnz >= ((int64_t)1 << 32))
        return fail("too many users, items or interactions", 0);
    memset(out, 0, sizeof(*out));
    unsigned long long *cnt = nullptr, *cursor = nullptr;
    int64_t *t_ptr = nullptr, *c_ptr = nullptr, *z_ptr = nullptr;
    int32_t *t_idx = nullptr, *t_idx2 = nullptr, *c_idx = nullptr;
    double *p_val = nullptr, *t_val = nullptr, *c_val = nullptr, *c_val2 = nullptr;
    uint64_t *k_in = nullptr, *k_out = nullptr;
    void *tmp = nullptr, *buf = nullptr;
    rocsparse_spmat_descr dA = nullptr, dB = nullptr, dC = nullptr, dD = nullptr;
    auto cleanup = [&](bool keep_out) {
        (void)hipFree(cnt); (void)hipFree(cursor); (void)hipFree(t_ptr); (void)hipFree(t_idx); (void)hipFree(t_idx2); (void)hipFree(p_val);
        (void)hipFree(t_val); (void)hipFree(c_val); (void)hipFree(c_val2); (void)hipFree(k_in); (void)hipFree(k_out);
        (void)hipFree(tmp); (void)hipFree(buf); (void)hipFree(c_idx); (void)hipFree(z_ptr);
        if (dA) rocsparse_destroy_spmat_descr(dA);
        if (dB) rocsparse_destroy_spmat_descr(dB);
        if (dC) rocsparse_destroy_spmat_descr(dC);
        if (dD) rocsparse_destroy_spmat_descr(dD);
        if (!keep_out) {
            (void)hipFree(c_ptr); (void)hipFree(out->w_idx); (void)hipFree(out->w_val); (void)hipFree(out->p_val32); (void)hipFree(out->deg);
            memset(out, 0, sizeof(*out));
        }
    };
#define FIT_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(false); return fail(#x, (int)e_); } } while (0)
#define FIT_SP(x) do { rocsparse_status s_ = (x); if (s_ != rocsparse_status_success) { cleanup(false); return fail(#x, (int)s_); } } while (0)
    const int64_t nz1 = nnz > 0 ? nnz : 1;
    FIT_HIP(hipMalloc((void **)&cnt, sizeof(unsigned long long) * (n_items + 1)));
    FIT_HIP(hipMalloc((void **)&cursor, sizeof(unsigned long long) * n_items));
    FIT_HIP(hipMalloc((void **)&t_ptr, sizeof(int64_t) * (n_items + 1)));
    FIT_HIP(hipMalloc((void **)&t_idx, sizeof(int32_t) * nz1));
    FIT_HIP(hipMalloc((void **)&p_val, sizeof(double) * nz1));
    FIT_HIP(hipMalloc((void **)&t_val, sizeof(double) * nz1));
    FIT_HIP(hipMalloc((void **)&out->p_val32, sizeof(float) * nz1));
    FIT_HIP(hipMalloc((void **)&out->deg, sizeof(int64_t) * n_items));
    FIT_HIP(hipMalloc((void **)&c_ptr, sizeof(int64_t) * (n_items + 1)));
    FIT_HIP(hipMalloc((void **)&z_ptr, sizeof(int64_t) * (n_items + 1)));                 // D = empty matrix
    FIT_HIP(hipMemset(z_ptr, 0, sizeof(int64_t) * (n_items + 1)));
    FIT_HIP(hipMemset(cnt, 0, sizeof(unsigned long long) * (n_items + 1)));
    FIT_HIP(hipMemset(cursor, 0, sizeof(unsigned long long) * n_items));
    // item degrees -> row pointers of A^T (exclusive scan)
    if (nnz > 0) hipLaunchKernelGGL(column_count_kernel, dim3((unsigned)std::min<int64_t>((nnz + 255) / 256, 1 << 20)), dim3(256), 0, 0, a_idx, nnz, cnt);
    size_t tmp_bytes = 0;
    auto cnt_in = rocprim::make_transform_iterator(cnt, to_i64());
    FIT_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, cnt_in, t_ptr, (int64_t)0, (size_t)(n_items + 1), rocprim::plus<int64_t>(), 0));
    FIT_HIP(hipMalloc(&tmp, tmp_bytes > 0 ? tmp_bytes : 16));
    FIT_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, cnt_in, t_ptr, (int64_t)0, (size_t)(n_items + 1), rocprim::plus<int64_t>(), 0));
    FIT_HIP(hipMemcpy(out->deg, cnt, sizeof(int64_t) * n_items, hipMemcpyDeviceToDevice));
    if (n_users > 0)
        hipLaunchKernelGGL(rp3_transpose_kernel, dim3((unsigned)((n_users + 255) / 256)), dim3(256), 0, 0, a_ptr, a_idx, n_users,
                           t_ptr, cursor, p_val, out->p_val32, t_idx, t_val);
    FIT_HIP(hipGetLastError());
    // users ascending inside every transposed row (the cursor above hands out positions in arrival order): the
    // numeric pass below adds in this order.  The values of a row are all 1 / deg(item): nothing to permute.
    if (nnz > 0) {
        FIT_HIP(hipMalloc((void **)&t_idx2, sizeof(int32_t) * nz1));
        size_t seg_bytes = 0;
        FIT_HIP(rocprim::segmented_radix_sort_keys(nullptr, seg_bytes, t_idx, t_idx2, (unsigned)nnz, (unsigned)n_items, t_ptr, t_ptr + 1, 0, 32));
        (void)hipFree(tmp); tmp = nullptr;
        FIT_HIP(hipMalloc(&tmp, seg_bytes > 0 ? seg_bytes : 16));
        FIT_HIP(rocprim::segmented_radix_sort_keys(tmp, seg_bytes, t_idx, t_idx2, (unsigned)nnz, (unsigned)n_items, t_ptr, t_ptr + 1, 0, 32));
        std::swap(t_idx, t_idx2);
    }

    rocsparse_handle sp = (rocsparse_handle)*rocsparse_cache;
    if (!sp) {
        FIT_SP(rocsparse_create_handle(&sp));
        *rocsparse_cache = sp;
    }
    FIT_SP(rocsparse_set_stream(sp, 0));
    // C = P_iu [items x users] . P_ui [users x items], fp64 like scipy's product
    FIT_SP(rocsparse_create_csr_descr(&dA, n_items, n_users, nnz, t_ptr, t_idx, t_val, rocsparse_indextype_i64, rocsparse_indextype_i32,
                                      rocsparse_index_base_zero, rocsparse_datatype_f64_r));
    FIT_SP(rocsparse_create_csr_descr(&dB, n_users, n_items, nnz, (void *)a_ptr, (void *)a_idx, p_val, rocsparse_indextype_i64,
                                      rocsparse_indextype_i32, rocsparse_index_base_zero, rocsparse_datatype_f64_r));
    FIT_SP(rocsparse_create_csr_descr(&dD, n_items, n_items, 0, z_ptr, nullptr, nullptr, rocsparse_indextype_i64, rocsparse_indextype_i32,
                                      rocsparse_index_base_zero, rocsparse_datatype_f64_r));
    FIT_SP(rocsparse_create_csr_descr(&dC, n_items, n_items, 0, c_ptr, nullptr, nullptr, rocsparse_indextype_i64, rocsparse_indextype_i32,
                                      rocsparse_index_base_zero, rocsparse_datatype_f64_r));
    const double alpha = 1.0;
    size_t buf_bytes = 0;
    FIT_SP(rocsparse_spgemm(sp, rocsparse_operation_none, rocsparse_operation_none, &alpha, dA, dB, nullptr, dD, dC, rocsparse_datatype_f64_r,
                            rocsparse_spgemm_alg_default, rocsparse_spgemm_stage_buffer_size, &buf_bytes, nullptr));
    FIT_HIP(hipMalloc(&buf, buf_bytes > 0 ? buf_bytes : 16));
    FIT_SP(rocsparse_spgemm(sp, rocsparse_operation_none, rocsparse_operation_none, &alpha, dA, dB, nullptr, dD, dC, rocsparse_datatype_f64_r,
                            rocsparse_spgemm_alg_default, rocsparse_spgemm_stage_nnz, &buf_bytes, buf));
    int64_t rows = 0, cols = 0, w_nnz = 0;
    FIT_SP(rocsparse_spmat_get_size(dC, &rows, &cols, &w_nnz));
    const int64_t wz1 = w_nnz > 0 ? w_nnz : 1;
    FIT_HIP(hipMalloc((void **)&c_idx, sizeof(int32_t) * wz1));
    FIT_HIP(hipMalloc((void **)&c_val, sizeof(double) * wz1));
    FIT_SP(rocsparse_csr_set_pointers(dC, c_ptr, c_idx, c_val));
    FIT_SP(rocsparse_spgemm(sp, rocsparse_operation_none, rocsparse_operation_none, &alpha, dA, dB, nullptr, dD, dC, rocsparse_datatype_f64_r,
                            rocsparse_spgemm_alg_default, rocsparse_spgemm_stage_compute, &buf_bytes, buf));
    FIT_HIP(hipDeviceSynchronize());
    // columns ascending inside every row: one radix sort of (row, column) keys carrying the values
    FIT_HIP(hipMalloc((void **)&k_in, sizeof(uint64_t) * wz1));
    FIT_HIP(hipMalloc((void **)&k_out, sizeof(uint64_t) * wz1));
    FIT_HIP(hipMalloc((void **)&c_val2, sizeof(double) * wz1));
    FIT_HIP(hipMalloc((void **)&out->w_idx, sizeof(int32_t) * wz1));
    FIT_HIP(hipMalloc((void **)&out->w_val, sizeof(float) * wz1));
    if (w_nnz > 0) {
        hipLaunchKernelGGL(rp3_keys_kernel, dim3((unsigned)n_items), dim3(256), 0, 0, c_ptr, c_idx, n_items, k_in);
        int row_bits = 1;
        while (((int64_t)1 << row_bits) < n_items) ++row_bits;
        size_t sort_bytes = 0;
        FIT_HIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, k_in, k_out, c_val, c_val2, (size_t)w_nnz, 0, 32 + row_bits, 0));
        (void)hipFree(tmp); tmp = nullptr;
        FIT_HIP(hipMalloc(&tmp, sort_bytes > 0 ? sort_bytes : 16));
        FIT_HIP(rocprim::radix_sort_pairs(tmp, sort_bytes, k_in, k_out, c_val, c_val2, (size_t)w_nnz, 0, 32 + row_bits, 0));
        // rocSPARSE gave the pattern; its values (summed in hash order) are replaced by the ordered sums
        hipLaunchKernelGGL(rp3_numeric_kernel, dim3((unsigned)std::min<int64_t>(n_items, 8192)), dim3(64), 0, 0, t_ptr, t_idx, a_ptr, a_idx,
                           c_ptr, k_out, c_val2, n_items);
        hipLaunchKernelGGL(rp3_unpack_kernel, dim3((unsigned)std::min<int64_t>((w_nnz + 255) / 256, 1 << 20)), dim3(256), 0, 0, k_out, c_val2, w_nnz, out->w_idx,
                           out->w_val);
    }
    FIT_HIP(hipGetLastError());
    FIT_HIP(hipDeviceSynchronize());
    out->w_ptr = c_ptr;
    out->w_nnz = w_nnz;
    cleanup(true);
    return 0;
#undef FIT_HIP
#undef FIT_SP
}

void rp3_free(Rp3Arrays *a)
{
    (void)hipFree(a->w_ptr); (void)hipFree(a->w_idx); (void)hipFree(a->w_val); (void)hipFree(a->p_val32); (void)hipFree(a->deg);
    memset(a, 0, sizeof(*a));
}

void rp3_release(void **rocsparse_cache)
{
    if (*rocsparse_cache) { rocsparse_destroy_handle((rocsparse_handle)*rocsparse_cache); *rocsparse_cache = nullptr; }
}

}  // namespace ifr
