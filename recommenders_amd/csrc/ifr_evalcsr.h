// ifr_evalcsr.h -- launch interface of the evaluation-CSR builder (ifr_evalcsr.hip) used by ifr_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ifr {

struct RawCsr {                  // one split as the reference's Dataset.csr_matrix holds it (device pointers)
    const int64_t *ptr;          // null: no such split
    const int32_t *idx;
    const void *val;             // ratings: float64 (f64 != 0) or fp32
    int64_t rows;
    int f64;
};

struct EvalCsrParams {
    RawCsr train, val, test;
    double thr;                  // rating_threshold
    int n_items;                 // catalogue: items >= n_items are clipped
    int64_t n_rows, row0;        // users [row0, row0 + n_rows) -> output rows [0, n_rows)
    int64_t *cnt_truth, *cnt_excl;   // [n_rows + 1]: counts, then (after the scan) the row pointers
    int32_t *run_a;              // [n_rows]: kept train entries per row (only with a validation split), else null
    uint8_t *has_pos;            // [n_rows]
    int *max_truth_row;          // [1]
    // fill sweep
    const int64_t *tr_ptr, *ex_ptr;
    int32_t *tr_idx, *ex_idx, *ex_tmp;
};

size_t eval_csr_scan_bytes(int64_t n_rows);
int launch_eval_csr_count(const EvalCsrParams &p, void *scan_tmp, size_t scan_bytes, hipStream_t s);
int launch_eval_csr_fill(const EvalCsrParams &p, hipStream_t s);

}  // namespace ifr
