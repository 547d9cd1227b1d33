// ifr_fit.h -- model build steps on the device (ifr_fit.hip), called from ifr_api.hip.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace ifr {

// EASE: d_ptr/d_idx/d_val = X (device CSR, columns strictly ascending); M [n_items][ldm] fp32 (device) receives B -- or,
// when M64 is not null, M64 [n_items][ldm] receives it in fp64; B_host (fp64 [n_items][n_items], may be null) a copy.
// rocblas_cache keeps the library handle between calls.
int ease_build(const int64_t *d_ptr, const int32_t *d_idx, const float *d_val, int64_t n_users, int64_t n_items, double lamb,
               int posB, float *M, double *M64, int64_t ldm, double *B_host, void **rocblas_cache, char *err, size_t errlen);
void ease_release(void **rocblas_cache);

// RP3-beta item-item form from the binarised train matrix A (device CSR, columns strictly ascending).
// Device arrays owned by the caller afterwards (rp3_free): W = P_iu . P_ui as CSR with ascending columns
// (fp32 values rounded from the fp64 product), the fp32 values of P_ui (same pattern as A), item degrees.
struct Rp3Arrays {
    int64_t *w_ptr;
    int32_t *w_idx;
    float *w_val;
    int64_t w_nnz;
    float *p_val32;
    int64_t *deg;
};
int rp3_build(const int64_t *a_ptr, const int32_t *a_idx, int64_t n_users, int64_t n_items, int64_t nnz, void **rocsparse_cache,
              Rp3Arrays *out, char *err, size_t errlen);
void rp3_free(Rp3Arrays *a);
void rp3_release(void **rocsparse_cache);

}  // namespace ifr
