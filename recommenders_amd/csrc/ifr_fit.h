// ifr_fit.h -- model build steps on the device (ifr_fit.hip), called from ifr_api.hip.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace ifr {

// EASE: d_ptr/d_idx/d_val = X (device CSR, columns strictly ascending); M [n_items][ldm] fp32 (device) receives B;
// B_host (fp64 [n_items][n_items], may be null) a copy.  rocblas_cache keeps the library handle between calls.
int ease_build(const int64_t *d_ptr, const int32_t *d_idx, const float *d_val, int64_t n_users, int64_t n_items, double lamb,
               int posB, float *M, int64_t ldm, double *B_host, void **rocblas_cache, char *err, size_t errlen);
void ease_release(void **rocblas_cache);

}  // namespace ifr
