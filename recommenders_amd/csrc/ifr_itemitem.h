// ifr_itemitem.h -- launch interface of the item-item kernels (ifr_itemitem.hip) used by ifr_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ifr {

struct ItemItemParams {
    const float *M;              // dense: [n_items][ldm]
    const double *M64;           // dense, fp64 option: [n_items][ldm] (then M is null)
    int64_t ldm;                 //        row stride, multiple of II_PASS
    const int64_t *cptr;         // sparse: [n_items][n_chunks + 1] entry positions per item chunk
    const int32_t *m_idx;        //         column of every entry
    const float *m_val;
    const uint2 *m_pair;         //         the same entries as interleaved (column, value bits) pairs (top-K kernel)
    int64_t m_nnz;               //         entries in m_idx / m_val
    int fx_shift;                //         fixed-point scale 2^fx_shift
    int dbg;                     // dev-only ablation bits (handle option debug_mode)
    const void *m_win;           //         [n_items][n_win][win_words] 16-byte window words (launch_build_win_words), or null
    int win_words;               //         1, 2 or 3
    const uint32_t *cptr_w;      //         [n_items][n_win + 1] entry offsets per 1024-column window (wave-per-user kernel), or null
    int n_win;
    int32_t *long_list;          //         users the wave-per-user kernel hands over to the workgroup kernel [n], and their count
    int *long_cnt;
    int sp_cap;                  //         CSR top-K: candidate buffer entries in use (item_sparse_min_capacity() .. item_sparse_pass_capacity())
    const float *pen;            // [>= n_chunks * chunk] column scale or null (CSR: already folded into m_val,
                                 // kept for the 0 * col_scale = NaN rule of exact zeros)
    const int64_t *h_ptr;        // history CSR
    const int32_t *h_idx;
    const float *h_val;
    int64_t h_rows;
    const int32_t *user_ids;
    int64_t n;
    int n_items;
    int n_chunks;
    const int64_t *ex_ptr;       // raw exclusion CSR (may be null)
    const int32_t *ex_idx;
    int64_t ex_rows;
    int K, KL;
    float *part_scores;          // [n][n_chunks][KL]   (TOPK)
    double *part_scores64;       // ... of the fp64 option (then part_scores is null): the lists are ranked on the fp64 sums
    int32_t *part_ids;
    float *out;                  // [n][ld]             (dense rows)
    double *out64;               // [n][ld]             (dense rows of the fp64 option, instead of out)
    int64_t ld;
};

int launch_item_item(const ItemItemParams &p, bool sparse, bool topk, hipStream_t stream);
// fp64 option: merge the per-chunk lists (double scores) of every user into the final top K, ranked on the doubles
int launch_merge_topk_f64(const double *part_scores, const int32_t *part_ids, int64_t n, int n_lists, int KL, int K, int32_t *out_ids,
                          float *out_scores, hipStream_t stream);
int launch_item_sparse_topk(const ItemItemParams &p, int32_t *out_ids, float *out_scores, hipStream_t stream);
int item_sparse_wave_windows(int n_items);
int launch_build_win_words(const uint32_t *cw, const int64_t *cptr, int n_chunks, const uint2 *pair, int64_t n_rows, int n_win, int words, void *out, hipStream_t s);
int item_sparse_wave_stride(int n_win);      // entries per row of the per-window offset table
int launch_count_long_segments(const uint32_t *cw, int64_t n_rows, int n_win, unsigned long long *count, hipStream_t s);
int launch_build_win_ptr(const int64_t *cptr, int n_chunks, const int32_t *idx, int64_t n_rows, int n_win, uint32_t *cw, hipStream_t s);
int item_sparse_pass_capacity();
int item_sparse_min_capacity();
int item_sparse_max_topk();
int item_sparse_read_phases(unsigned long long *out16);   // dev build only (make ABLATE=1), else -1
int item_sparse_max_chunks();
int item_csr_pad();             // entries the CSR index / value arrays must be allocated past nnz
int item_item_chunk(bool sparse);
int item_item_chunk_f64();      // dense fp64 matrix: columns per workgroup
int item_item_pass();
int launch_validate_csr(const int64_t *ptr, const int32_t *idx, int64_t n_rows, int n_cols, int need_sorted, int *bad,
                        hipStream_t s);
int launch_build_chunk_ptr(const int64_t *ptr, const int32_t *idx, int64_t n_rows, int n_chunks, int chunk, int64_t *cptr,
                           hipStream_t s);
int launch_scale_csr_values(float *val, const int32_t *idx, int64_t nnz, const float *col_scale, hipStream_t s);
int launch_interleave_csr(const int32_t *idx, const float *val, int64_t n, uint2 *pair, hipStream_t s);
int launch_absmax(const float *x, int64_t n, uint32_t *out, hipStream_t s);
int launch_row_l1_max(const int64_t *ptr, const float *val, int64_t n_rows, uint32_t *out, hipStream_t s);

}  // namespace ifr
