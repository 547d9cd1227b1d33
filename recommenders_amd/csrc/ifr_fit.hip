// ifr_fit.hip -- item-item model build steps on the device.
//
// EASE closed form, cornac/models/ease/recom_ease.py:75-95:
//     G = X^T X + lamb I          (X = train_set.matrix, CSR [users][items])
//     P = G^-1
//     B = P / (-diag P)  column-wise,  diag B = 0,  posB: negative entries -> 0
// G and P are fp64 like the reference's.  The Gram matrix is a scatter of every user's item pairs
// (nnz_u^2 / 2 fp64 atomics into the stored triangle: HBM-bound, no GEMM to shape: a dense X^T X
// would do users / nnz_u times the flops); the inverse is a plain library factorisation
// (rocSOLVER Cholesky + solve against the identity, 64-bit index API), the final scaling a
// streaming kernel that writes the fp32 matrix the scoring kernels read.
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <stdint.h>
#include <stdio.h>

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
__global__ __launch_bounds__(FIT_BLOCK) void ease_scale_kernel(double *S, const double *d, int64_t n, int posB, float *M, int64_t ldm)
{
    const int64_t j = blockIdx.y;
    const int64_t c = (int64_t)blockIdx.x * FIT_BLOCK + threadIdx.x;
    if (c >= n) return;
    double v = S[j * n + c] / (-d[c]);
    if (j == c) v = 0.0;
    if (posB && v < 0.0) v = 0.0;
    S[j * n + c] = v;
    M[j * ldm + c] = (float)v;
}

static int fit_fail(char *err, size_t errlen, const char *what, int code)
{
    if (err && errlen) snprintf(err, errlen, "ifr_fit_ease: %s (%d)", what, code);
    return -1;
}

int ease_build(const int64_t *d_ptr, const int32_t *d_idx, const float *d_val, int64_t n_users, int64_t n_items, double lamb,
               int posB, float *M, int64_t ldm, double *B_host, void **rocblas_cache, char *err, size_t errlen)
{
    const int64_t n = n_items;
    if (n_users > 0x7fffffff || n > 65535LL * FIT_BLOCK) return fit_fail(err, errlen, "too many users or items for one launch", 0);
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
    hipLaunchKernelGGL(ease_scale_kernel, dim3(nb, (unsigned)n), dim3(FIT_BLOCK), 0, 0, S, diag, n, posB, M, ldm);
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

}  // namespace ifr
