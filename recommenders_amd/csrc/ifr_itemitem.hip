// ifr_itemitem.hip -- item-item models (EASE, RP3-beta):
//     s_u = (sum_{j in hist(u)} x_uj * M[j, :]) * col_scale
//
// Reference: EASE.score  cornac/models/ease/recom_ease.py:123-124  (X[u,:].dot(B): sparse row x dense B)
//            RP3-beta     cornac/models/rp3_beta/graph_recommender.py:97-117, in item-item form:
//                         P3[u, items] = P_ui[u,:] @ (P_iu @ P_ui), times 1/deg^beta.
//
// HBM/L2-bound row gathers: a user streams nnz_u rows of M.  There is no GEMM to shape (a dense X.B
// contraction would do I/nnz_u ~ 1000x the flops), so no MFMA here.
//
//   dense M  [n_items][ldm] fp32: coalesced 16-B row reads, one fp32 fma chain per (user, item) in
//            history order (the order of the user's CSR row), scores held in registers.
//   sparse M (CSR, columns ascending, blocked by item chunk at set time): every nonzero of the
//            user's history rows is scattered into a per-workgroup LDS accumulator.  The accumulator
//            is 64-bit fixed point (product in exact fp64, scaled by 2^shift, rounded to nearest
//            even, added with an integer LDS atomic), so the sum does not depend on the order in
//            which the waves arrive: results are bit-reproducible and a CPU checker can restate them.
//            col_scale is folded into the stored values at set time (one fp32 rounding per entry).
//
// Both end in the same epilogue: column scale, exclusion bitmap in LDS, block-wide top-K of the
// chunk (score descending, item index ascending), partial lists merged by merge_topk_kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "ifr_itemitem.h"

namespace ifr {

constexpr int II_BLOCK = 256;
constexpr int II_PASS = 1024;                 // items per pass: 256 threads x float4
constexpr int II_HIST = 512;                  // history entries staged per round
constexpr int II_NONE = 0x7fffffff;
constexpr int II_DENSE_PASSES = 8;            // dense M: 8192 items per workgroup
constexpr int II_SPARSE_PASSES = 4;           // sparse M: 4096 items per workgroup (32 KB of int64 LDS)
constexpr int II_F64_PASSES = 4;              // dense fp64 M: 4096 items per workgroup (fp64 accumulators: twice the registers)
constexpr int II_GROUP = 16;                  // lanes that share one sparse row segment
constexpr int II_CSR_PAD = 64;                // entries allocated past the end of the CSR arrays (over-read, never used)

typedef float f32x4 __attribute__((ext_vector_type(4)));


__device__ __forceinline__ bool ii_beats(float s, int id, float s2, int id2)
{
    return (s > s2) || (s == s2 && id < id2);
}

// F64 (dense M only): M is fp64 and every (user, item) sum is one fp64 fma chain in history order, like the reference's
// X[u, :].dot(B) in float64 (recom_ease.py:123-124); the column scale is applied in fp64, the sum is written as fp64
// (out64) or rounded once to fp32 for the ranking epilogue.
template <int PASSES, bool SPARSE, bool TOPK, bool F64 = false>
__global__ __launch_bounds__(II_BLOCK, 3) void item_item_kernel(ItemItemParams p)
{
    constexpr int CHUNK = PASSES * II_PASS;
    // sparse: int64 accumulators, later reused for the fp32 scores of the epilogue
    __shared__ __attribute__((aligned(16))) unsigned char buf[SPARSE ? CHUNK * 8 : (TOPK ? CHUNK * (F64 ? 8 : 4) : 16)];
    __shared__ uint32_t bm[TOPK ? CHUNK / 32 : 1];
    __shared__ int hj[II_HIST];
    __shared__ float hx[II_HIST];
    __shared__ int64_t sb[SPARSE ? II_HIST : 1], se[SPARSE ? II_HIST : 1];
    __shared__ float red_s[II_BLOCK / 64];
    __shared__ double red_d[F64 && TOPK ? II_BLOCK / 64 : 1];
    __shared__ int red_i[II_BLOCK / 64];
    __shared__ int win_id;
    float *sc = reinterpret_cast<float *>(buf);
    unsigned long long *acc64 = reinterpret_cast<unsigned long long *>(buf);

    const int tid = threadIdx.x;
    const int chunk = blockIdx.y;                        // x = user: co-resident workgroups share an item chunk
    const int64_t i = blockIdx.x;
    const int u = p.user_ids[i];
    const int c0 = chunk * CHUNK;

    int64_t hb = 0, he = 0;
    if (u >= 0 && u < p.h_rows) { hb = p.h_ptr[u]; he = p.h_ptr[u + 1]; }

    f32x4 acc[PASSES];
#pragma unroll
    for (int q = 0; q < PASSES; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int col0 = c0 + 4 * tid;                       // first of this thread's 4 columns in pass 0
    const int npass = min(PASSES, (p.n_items - c0 + II_PASS - 1) / II_PASS);   // passes that hold items (wave-uniform)

    if (SPARSE) {
        for (int w = tid; w < CHUNK; w += II_BLOCK) acc64[w] = 0ull;
        const int g = tid / II_GROUP, l = tid % II_GROUP;
        for (int64_t base = hb; base < he; base += II_HIST) {
            const int m = (int)min((int64_t)II_HIST, he - base);
            __syncthreads();
            for (int e = tid; e < m; e += II_BLOCK) {
                const int j = p.h_idx[base + e];
                hj[e] = j;
                hx[e] = p.h_val[base + e];
                const int64_t *cp = p.cptr + (int64_t)j * (p.n_chunks + 1) + chunk;
                sb[e] = cp[0];
                se[e] = cp[1];
            }
            __syncthreads();
            for (int e = g; e < m; e += II_BLOCK / II_GROUP) {
                const double x = (double)hx[e];
                const int64_t q1 = se[e];
                for (int64_t q = sb[e] + l; q < q1; q += II_GROUP) {
                    const int c = p.m_idx[q] - c0;
                    const double pr = x * (double)p.m_val[q];                  // exact: 24 x 24 bits
                    const long long v = __double2ll_rn(ldexp(pr, p.fx_shift));  // nearest even
                    atomicAdd(&acc64[c], (unsigned long long)v);
                }
            }
        }
        __syncthreads();
        if (!TOPK && p.out64 != nullptr) {
            // fp64 rows of a CSR matrix: the exact fixed-point sum scaled back (|sum| < 2^52: nothing is rounded); a zero sum
            // still meets the column scale (0 * inf = NaN like the reference)
#pragma unroll
            for (int q = 0; q < PASSES; ++q) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int col = col0 + q * II_PASS + v;
                    if (col >= p.n_items) continue;
                    double sd = ldexp((double)(long long)acc64[q * II_PASS + 4 * tid + v], -p.fx_shift);
                    if (sd == 0.0) sd = p.pen != nullptr ? 0.0 * (double)p.pen[col] : 0.0;
                    p.out64[i * p.ld + col] = sd;
                }
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const long long a = (long long)acc64[q * II_PASS + 4 * tid + v];
                acc[q][v] = (float)ldexp((double)a, -p.fx_shift);
            }
        }
        __syncthreads();                                 // buf is about to be rewritten as fp32 scores
    } else if (F64) {
        double a64[PASSES][4];
#pragma unroll
        for (int q = 0; q < PASSES; ++q)
#pragma unroll
            for (int v = 0; v < 4; ++v) a64[q][v] = 0.0;
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        for (int64_t base = hb; base < he; base += II_HIST) {
            const int m = (int)min((int64_t)II_HIST, he - base);
            __syncthreads();
            for (int e = tid; e < m; e += II_BLOCK) { hj[e] = p.h_idx[base + e]; hx[e] = p.h_val[base + e]; }
            __syncthreads();
#pragma unroll 1
            for (int e = 0; e < m; ++e) {
                const double x = (double)hx[e];
                const double *row = p.M64 + (int64_t)hj[e] * p.ldm + col0;
                f64x2 r[PASSES][2];
#pragma unroll
                for (int q = 0; q < PASSES; ++q) {
                    if (q < npass) {
                        r[q][0] = *reinterpret_cast<const f64x2 *>(row + q * II_PASS);
                        r[q][1] = *reinterpret_cast<const f64x2 *>(row + q * II_PASS + 2);
                    }
                }
#pragma unroll
                for (int q = 0; q < PASSES; ++q) {
                    if (q < npass) {
                        a64[q][0] = fma(x, r[q][0][0], a64[q][0]);
                        a64[q][1] = fma(x, r[q][0][1], a64[q][1]);
                        a64[q][2] = fma(x, r[q][1][0], a64[q][2]);
                        a64[q][3] = fma(x, r[q][1][1], a64[q][3]);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
            if (q < npass) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int col = col0 + q * II_PASS + v;
                    double s = a64[q][v];
                    if (p.pen != nullptr) s *= (double)p.pen[col];
                    if (p.out64 != nullptr && col < p.n_items) p.out64[i * p.ld + col] = s;      // (TOPK: the rows as well, one scoring pass for both)
                    a64[q][v] = s;
                    acc[q][v] = (float)s;
                }
            }
        }
        if (!TOPK && p.out64 != nullptr) return;
        if (TOPK) {
            // ---- the epilogue below, on the fp64 sums themselves: the reference orders its float64 scores
            // (recom_ease.py:123-124 -> recommender.py:643-652), so nothing is rounded before it is ranked
            double *const sd = reinterpret_cast<double *>(buf);
            for (int w = tid; w < CHUNK / 32; w += II_BLOCK) bm[w] = 0;
            __syncthreads();
            if (p.ex_ptr != nullptr && u >= 0 && u < p.ex_rows) {
                const int64_t eb = p.ex_ptr[u], ee = p.ex_ptr[u + 1];
                int64_t lo = eb, hi = ee;
                while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (p.ex_idx[mid] < c0) lo = mid + 1; else hi = mid; }
                for (int64_t e = lo + tid; e < ee; e += II_BLOCK) {
                    const int it = p.ex_idx[e] - c0;
                    if (it >= CHUNK) break;
                    atomicOr(&bm[it >> 5], 1u << (it & 31));
                }
            }
            __syncthreads();
            const double dnan = __longlong_as_double(0x7ff8000000000000LL);
            auto beats64 = [](double s, int id, double s2, int id2) { return (s > s2) || (s == s2 && id < id2); };
            double lmax = -INFINITY;
            int lid = II_NONE;
#pragma unroll
            for (int q = 0; q < PASSES; ++q) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int loc = q * II_PASS + 4 * tid + v;
                    const int col = c0 + loc;
                    double s = q < npass ? a64[q][v] + 0.0 : 0.0;                 // + 0: -0 ranks as +0
                    const bool dead = col >= p.n_items || ((bm[loc >> 5] >> (loc & 31)) & 1u) || !(s == s);
                    if (dead) s = dnan;
                    sd[loc] = s;
                    if (!dead && beats64(s, col, lmax, lid)) { lmax = s; lid = col; }
                }
            }
            const int lane = tid & 63, wave = tid >> 6;
            const int64_t obase = (i * p.n_chunks + chunk) * p.KL;
            for (int r = 0; r < p.KL; ++r) {
                double bs = lmax;
                int bi = lid;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const double os = __shfl_xor(bs, off);
                    const int oi = __shfl_xor(bi, off);
                    if (oi != II_NONE && (bi == II_NONE || beats64(os, oi, bs, bi))) { bs = os; bi = oi; }
                }
                if (lane == 0) { red_d[wave] = bs; red_i[wave] = bi; }
                __syncthreads();
                if (tid == 0) {
                    double ws = red_d[0];
                    int wi = red_i[0];
                    for (int w = 1; w < II_BLOCK / 64; ++w)
                        if (red_i[w] != II_NONE && (wi == II_NONE || beats64(red_d[w], red_i[w], ws, wi))) { ws = red_d[w]; wi = red_i[w]; }
                    win_id = wi;
                    p.part_scores64[obase + r] = wi == II_NONE ? -INFINITY : ws;
                    p.part_ids[obase + r] = wi;
                }
                __syncthreads();
                const int wi = win_id;
                if (wi == II_NONE || r >= p.K - 1) {
                    if (tid == 0)
                        for (int rr = r + 1; rr < p.KL; ++rr) { p.part_scores64[obase + rr] = -INFINITY; p.part_ids[obase + rr] = II_NONE; }
                    break;
                }
                if (wi == lid) {
                    sd[wi - c0] = dnan;
                    lmax = -INFINITY; lid = II_NONE;
#pragma unroll
                    for (int q = 0; q < PASSES; ++q) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int loc = q * II_PASS + 4 * tid + v;
                            const double s = sd[loc];
                            if (s == s && beats64(s, c0 + loc, lmax, lid)) { lmax = s; lid = c0 + loc; }
                        }
                    }
                }
            }
            return;
        }
    } else {
        for (int64_t base = hb; base < he; base += II_HIST) {
            const int m = (int)min((int64_t)II_HIST, he - base);
            __syncthreads();
            for (int e = tid; e < m; e += II_BLOCK) { hj[e] = p.h_idx[base + e]; hx[e] = p.h_val[base + e]; }
            __syncthreads();
            // two history rows per iteration: 16 independent 16-B loads in flight per thread; the
            // fma chain of every column still runs in history order.  A chunk that lies fully
            // inside the catalogue takes the branch-free body so that the loads batch.
            auto rows2 = [&](int e, auto full) {
                const float x0 = hx[e], x1 = hx[e + 1];
                const float *row0 = p.M + (int64_t)hj[e] * p.ldm + col0;
                const float *row1 = p.M + (int64_t)hj[e + 1] * p.ldm + col0;
                f32x4 r0[PASSES], r1[PASSES];
#pragma unroll
                for (int q = 0; q < PASSES; ++q) {
                    if (decltype(full)::value || q < npass) {
                        r0[q] = *reinterpret_cast<const f32x4 *>(row0 + q * II_PASS);
                        r1[q] = *reinterpret_cast<const f32x4 *>(row1 + q * II_PASS);
                    }
                }
#pragma unroll
                for (int q = 0; q < PASSES; ++q) {
                    if (decltype(full)::value || q < npass) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc[q][v] = fmaf(x1, r1[q][v], fmaf(x0, r0[q][v], acc[q][v]));
                    }
                }
            };
            int e = 0;
            if (npass == PASSES) {
#pragma unroll 1
                for (; e + 1 < m; e += 2) rows2(e, std::true_type{});
            } else {
#pragma unroll 1
                for (; e + 1 < m; e += 2) rows2(e, std::false_type{});
            }
            if (e < m) {
                const float x = hx[e];
                const float *row = p.M + (int64_t)hj[e] * p.ldm + col0;
#pragma unroll
                for (int q = 0; q < PASSES; ++q) {
                    if (q < npass) {
                        const f32x4 r = *reinterpret_cast<const f32x4 *>(row + q * II_PASS);
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc[q][v] = fmaf(x, r[v], acc[q][v]);
                    }
                }
            }
        }
    }
    if (p.pen != nullptr && !F64) {
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
            if (q < npass) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(p.pen + col0 + q * II_PASS);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    // CSR values carry col_scale already; only an exact zero still meets it (0 * inf = NaN)
                    if (SPARSE) acc[q][v] = acc[q][v] == 0.0f ? acc[q][v] * w[v] : acc[q][v];
                    else acc[q][v] *= w[v];
                }
            }
        }
    }

    if (!TOPK || (!SPARSE && !F64 && p.out != nullptr)) {      // dense rows; in a TOPK launch: as well as the lists (one scoring pass)
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int col = col0 + q * II_PASS + v;
                if (col < p.n_items) p.out[i * p.ld + col] = acc[q][v];
            }
        }
        if (!TOPK) return;
    }

    // ---- exclusion bitmap of this chunk (bit set = not a candidate), then scores -> LDS
    for (int w = tid; w < CHUNK / 32; w += II_BLOCK) bm[w] = 0;
    __syncthreads();
    if (p.ex_ptr != nullptr && u >= 0 && u < p.ex_rows) {
        const int64_t eb = p.ex_ptr[u], ee = p.ex_ptr[u + 1];
        int64_t lo = eb, hi = ee;                        // first excluded item >= c0
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (p.ex_idx[mid] < c0) lo = mid + 1; else hi = mid; }
        for (int64_t e = lo + tid; e < ee; e += II_BLOCK) {
            const int it = p.ex_idx[e] - c0;
            if (it >= CHUNK) break;
            atomicOr(&bm[it >> 5], 1u << (it & 31));
        }
    }
    __syncthreads();
    float lmax = -INFINITY;                              // this thread's best remaining candidate
    int lid = II_NONE;
#pragma unroll
    for (int q = 0; q < PASSES; ++q) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int loc = q * II_PASS + 4 * tid + v;
            const int col = c0 + loc;
            float s = acc[q][v];
            const bool dead = col >= p.n_items || ((bm[loc >> 5] >> (loc & 31)) & 1u) || !(s == s);
            if (dead) s = __uint_as_float(0x7fc00000u);  // NaN: never a candidate
            sc[loc] = s;
            if (!dead && ii_beats(s, col, lmax, lid)) { lmax = s; lid = col; }
        }
    }
    // ---- K rounds of block-wide argmax (score desc, id asc); the winner's owner rescans its slots
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t obase = (i * p.n_chunks + chunk) * p.KL;
    for (int r = 0; r < p.KL; ++r) {
        float bs = lmax;
        int bi = lid;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float os = __shfl_xor(bs, off);
            const int oi = __shfl_xor(bi, off);
            if (oi != II_NONE && (bi == II_NONE || ii_beats(os, oi, bs, bi))) { bs = os; bi = oi; }
        }
        if (lane == 0) { red_s[wave] = bs; red_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float ws = red_s[0];
            int wi = red_i[0];
            for (int w = 1; w < II_BLOCK / 64; ++w)
                if (red_i[w] != II_NONE && (wi == II_NONE || ii_beats(red_s[w], red_i[w], ws, wi))) { ws = red_s[w]; wi = red_i[w]; }
            win_id = wi;
            p.part_scores[obase + r] = wi == II_NONE ? -INFINITY : ws;
            p.part_ids[obase + r] = wi;
        }
        __syncthreads();
        const int wi = win_id;
        if (wi == II_NONE || r >= p.K - 1) {             // nothing left / enough ranks: pad the rest
            if (tid == 0)
                for (int rr = r + 1; rr < p.KL; ++rr) { p.part_scores[obase + rr] = -INFINITY; p.part_ids[obase + rr] = II_NONE; }
            break;
        }
        if (wi == lid) {                                 // owner: retire the winner, find the next best
            sc[wi - c0] = __uint_as_float(0x7fc00000u);
            lmax = -INFINITY; lid = II_NONE;
#pragma unroll
            for (int q = 0; q < PASSES; ++q) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int loc = q * II_PASS + 4 * tid + v;
                    const float s = sc[loc];
                    if (s == s && ii_beats(s, c0 + loc, lmax, lid)) { lmax = s; lid = c0 + loc; }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// CSR M, top-K (round 3): one workgroup per user sweeps the catalogue in WINDOWS of 4096 columns -- the granularity of
// the per-row entry positions `cptr` built at set time -- and keeps ONE running top-K list.
//
//   accumulate : every history row's entries of the window (a contiguous segment: cptr[row][c] .. cptr[row][c + 1]) are
//                added into 4096 DIRECT-MAPPED fp64 accumulators in LDS: no hashing, no probing, no key compares.
//                The addend is the exact product x_uj * M'[j, c] * 2^shift rounded to the nearest-even INTEGER by one
//                fma against 1.5 * 2^52 (three VALU instructions from the loaded value to the addend; round 2 converted
//                to int64 in ~30).  The shift keeps every product and every partial sum below 2^51, so the fp64 adds
//                (native ds_add_f64) are EXACT: the sum does not depend on the order in which the waves arrive, is
//                bit-reproducible, and a CPU checker restates it with 64-bit integers (ifr_item_fixed_point_shift).
//   exclusions : the user's excluded columns of the window are stored as a poison value (-2^60) that no sum can lift.
//   scan       : every thread swaps its 16 accumulators for zero (one LDS exchange each: read and clear), so the table is
//                clean for the next window; a max tree + ONE compare against the list's K-th sum decides whether the
//                thread holds a candidate at all (after the first windows almost never).
//   select     : in the EXACT domain -- candidates are (sum, column) pairs ordered by sum descending, column ascending;
//                nothing is converted or rounded before it is ranked (only the K winners are scaled back and rounded to
//                fp32 at the end).  New candidates go behind a copy of the list and are ranked by counting.  While the
//                list is short, a lower bound of the window's K-th best comes from K rounds of a wave maximum over the
//                threads' best sums.  A window with more candidates than the buffer holds (mass ties) is ranked by K
//                rounds of an exact block-wide argmax over the register-resident sums instead.
//   zeros      : columns whose sum is exactly 0 (no history row touches them, or the entries cancel) score
//                0 * col_scale.  They only matter when the list ends up short or holds sums <= 0; only then a second
//                sweep takes, in column order, the zero columns that still rank.
// The work per user is |entries| + n_items / 64 wave-level slot visits: it follows the catalogue size as well as the
// user's nonzeros (round 2's hash passes followed the nonzeros only, at ~250 VALU instructions per 64 entries).
// ------------------------------------------------------------------------------------
constexpr int SP_CHUNK = II_SPARSE_PASSES * II_PASS;   // 4096 columns: granularity of cptr = one window
constexpr int SD_BLOCK = 256;                           // threads per user
constexpr int SD_SPT = SP_CHUNK / SD_BLOCK;             // 16 accumulators per thread
constexpr int SD_HIST = 256;                            // history rows staged per round
constexpr int SD_PRE = 3;                               // entries per lane prefetched one window ahead
constexpr int SD_EXCAP = 224;                           // exclusion row entries kept in LDS
constexpr int SD_CAND = 256;                            // candidate buffer: the new candidates that wait to be merged into the list
constexpr int SD_MAXK = 128;                            // longest list (LDS of a workgroup: 40 912 B, four workgroups per CU for every K)
constexpr int SD_FLOOR_BITS = 18;                       // bits of the short-list bound (sign + exponent + 6 of the mantissa)
constexpr int SD_LAZY = 16;                             // (+ K) pending candidates trigger a merge ...
constexpr int SD_LAZY_MAX = 48;                         // ... at most this many (measured at K = 50 ... 128: 16, 32, 48, 56, 16 + K)
constexpr double SD_MAGIC = 6755399441055744.0;         // 1.5 * 2^52: x + MAGIC - MAGIC = x rounded to the nearest-even integer, |x| < 2^51
constexpr double SD_POISON = -1152921504606846976.0;    // -2^60: an excluded column
constexpr double SD_LIVE = -288230376151711744.0;       // -2^58: sums above are real (|sum| < 2^52)

__device__ __forceinline__ bool sd_beats(double v, int c, double v2, int c2)
{
    return (v > v2) || (v == v2 && c < c2);
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() fences global memory as well, and its acquire side made
// the compiler drain every global load in flight (s_waitcnt vmcnt(0)) right after each barrier -- the loads this kernel
// issues a window ahead precisely so that they stay in flight across the barriers.  Nothing here communicates through
// global memory (inputs are read-only, a thread writes its own outputs at the end).
__device__ __forceinline__ void sd_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ double sd_wave_max(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}

#ifdef IFR_ABLATE
// dev build only (make ABLATE=1): cycles per phase of a window, summed over wave 0 of every workgroup
__device__ unsigned long long sd_stamps[16];
#define SD_T0() unsigned long long sd_t = __builtin_amdgcn_s_memtime(), sd_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; (void)sd_t
#define SD_STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); sd_acc[k] += t_ - sd_t; sd_t = t_; } while (0)
#else
#define SD_T0() do {} while (0)
#define SD_STAMP(k) do {} while (0)
#endif

template <bool IDX32, bool LIST>
__global__ __launch_bounds__(SD_BLOCK, 4) void item_sparse_topk_kernel(ItemItemParams p, int32_t *out_ids, float *out_scores)
{
    __shared__ __attribute__((aligned(16))) double tab[SP_CHUNK + 64];        // + a dummy slot per lane
    __shared__ float xf[SD_HIST];                        // history weights
    __shared__ int hj[SD_HIST];
    __shared__ int exl[SD_EXCAP];
    __shared__ double cand_v[SD_CAND];
    __shared__ int cand_c[SD_CAND];
    __shared__ double top_v[SD_MAXK];
    __shared__ int top_c[SD_MAXK];
    __shared__ double wred_v[SD_BLOCK / 64];
    __shared__ int wred_c[SD_BLOCK / 64];
    __shared__ uint32_t wred_n[SD_BLOCK / 64];
    __shared__ int s_nq, s_cnt, s_dead;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // direct mode: workgroup b ranks user b.  LIST mode: the wave-per-user kernel below handed over the
    // users it does not serve; the workgroups of a fixed-size grid take them in turns.
    const int64_t limit = LIST ? (int64_t)*p.long_cnt : p.n;
    int64_t it = blockIdx.x;
    if (it >= limit) return;
    do {
    const int64_t i = LIST ? (int64_t)p.long_list[it] : it;
    SD_T0();
    const int u = p.user_ids[i];
    const int K = p.K;
    const int nc1 = p.n_chunks + 1;
    const int cap = p.sp_cap;                             // candidate buffer entries in use (K ... SD_CAND; tests shrink it)
    // K > 64: a wave (64 lanes) cannot count K lanes above a trial bound.  Every wave then bounds its own quarter --
    // at least ceil(K / 4) of its lanes at or above -- and the workgroup takes the SMALLEST of the four: 4 ceil(K / 4) >= K
    // distinct columns reach it.  (K <= 64: every wave's K-th best is a bound by itself, the largest is taken.)
    const int lazy = min(SD_LAZY + K, SD_LAZY_MAX);     // pending candidates that trigger a merge while the list is full
    const bool split = K > 4;                             // (measured: splitting pays from K = 8 on, +2 % at K = 10, +8 % at K = 32; K = 64 unsplit ran slower than K = 100 split)
    const int Kw = split ? (K + SD_BLOCK / 64 - 1) / (SD_BLOCK / 64) : K;

    int64_t hb = 0, he = 0, eb = 0, ee = 0;
    if (u >= 0 && u < p.h_rows) { hb = p.h_ptr[u]; he = p.h_ptr[u + 1]; }
    if (p.ex_ptr != nullptr && u >= 0 && u < p.ex_rows) { eb = p.ex_ptr[u]; ee = p.ex_ptr[u + 1]; }
    const int64_t n_ex = ee - eb;
    const bool ex_lds = n_ex <= SD_EXCAP;
    const bool single = (he - hb) <= min(SD_HIST, SD_BLOCK);      // one staging round, and a thread per history row
    const int m1 = single ? (int)(he - hb) : 0;

    for (int w = tid; w < SP_CHUNK + 64; w += SD_BLOCK) tab[w] = 0.0;
    if (ex_lds) for (int w = tid; w < (int)n_ex; w += SD_BLOCK) exl[w] = p.ex_idx[eb + w];
    if (tid == 0) { s_nq = 0; s_cnt = 0; s_dead = 0; }
    auto stage = [&](int64_t base, int m) {
        for (int e = tid; e < m; e += SD_BLOCK) { hj[e] = p.h_idx[base + e]; xf[e] = p.h_val[base + e]; }
    };
    if (single) stage(hb, m1);
    sd_barrier();

    // first exclusion entry with column >= col
    auto ex_lower_bound = [&](int col) -> int64_t {
        int64_t lo = 0, hi = n_ex;
        if (ex_lds) { while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (exl[mid] < col) lo = mid + 1; else hi = mid; } }
        else { while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (p.ex_idx[eb + mid] < col) lo = mid + 1; else hi = mid; } }
        return lo;
    };
    // the user's exclusions with colA <= col < colB become poison slots (stores race with the adds of the same window:
    // either order leaves the slot far below every real sum)
    auto poison = [&](int colA, int colB) __attribute__((always_inline)) {
        if (n_ex == 0) return;
        if (ex_lds) {
            for (int w = tid; w < (int)n_ex; w += SD_BLOCK) { const int c = exl[w]; if (c >= colA && c < colB) tab[c - colA] = SD_POISON; }
        } else {
            for (int64_t w = ex_lower_bound(colA) + tid; w < n_ex; w += SD_BLOCK) {
                const int c = p.ex_idx[eb + w];
                if (c >= colB) break;
                tab[c - colA] = SD_POISON;
            }
        }
    };
    // rows [0, m) of the staged history: L lanes share a row's segment of window c
    auto accumulate_rows = [&](int m, int c, int colA) __attribute__((always_inline)) {
        int L = SD_BLOCK;
        while (L > 1 && SD_BLOCK / L < m) L >>= 1;        // largest power of two with (SD_BLOCK / L) >= m rows at once
        const int rows_at_once = SD_BLOCK / L;
        const int l = tid & (L - 1);
        for (int e = tid / L; e < m; e += rows_at_once) {
            const int64_t *cp = p.cptr + (int64_t)hj[e] * nc1 + c;
            const int64_t q0 = cp[0], q1 = cp[1];
            const double x = ldexp((double)xf[e], p.fx_shift);      // history weight * 2^shift
            for (int64_t q = q0 + l; q < q1; q += L) {
                const uint2 e2 = p.m_pair[q];
                const double r = fma(x, (double)__uint_as_float(e2.y), SD_MAGIC) - SD_MAGIC;     // exact product, nearest-even integer
                atomicAdd(&tab[(int)e2.x - colA], r);
            }
        }
    };
    auto accumulate = [&](int c, int colA) __attribute__((always_inline)) {
        if (single) accumulate_rows(m1, c, colA);
        else {
            for (int64_t base = hb; base < he; base += SD_HIST) {
                const int m = (int)min((int64_t)SD_HIST, he - base);
                sd_barrier();
                stage(base, m);
                sd_barrier();
                accumulate_rows(m, c, colA);
            }
        }
    };
    // the list (top_*[0, cnt), sorted) + the new candidates cand[0, s_nq)  ->  the new list.  Call after a barrier.
    // Ranks by counting, but only against what is NOT sorted yet: a list entry keeps its position plus the number of
    // candidates that beat it; a candidate's rank is its position in the list (binary search) plus the number of
    // candidates that beat it.  (Counting every entry against every entry cost (K + P)^2 compares for P candidates:
    // at K = 100 the merges were 3/4 of the kernel's instructions.)  List entries sit on the low threads, candidates
    // on the high ones, so that the waves share the work.
    // (check_ex: the new candidates have not been looked up in the user's exclusion row yet -- the entry-driven path
    // defers that to here, where one lookup per candidate runs for all of them at once)
    auto rank_merge = [&](bool check_ex) __attribute__((always_inline)) {
        const int total = s_nq, cnt = s_cnt;
        if (check_ex && n_ex != 0) {
            if (tid < total) {
                const int col = cand_c[tid];
                const int64_t w = ex_lower_bound(col);
                if (w < n_ex && (ex_lds ? exl[w] : p.ex_idx[eb + w]) == col) { cand_v[tid] = -INFINITY; atomicAdd(&s_dead, 1); }
            }
            sd_barrier();
        }
        int r_list = K;
        double lv = 0.0;
        int lc = 0;
        if (tid < cnt) {
            lv = top_v[tid]; lc = top_c[tid];
            int b = 0;
            for (int x = 0; x < total; ++x) b += sd_beats(cand_v[x], cand_c[x], lv, lc) ? 1 : 0;
            r_list = tid + b;
        }
        int r_new = K;
        double nv = 0.0;
        int ncol = 0;
        const int e = SD_BLOCK - 1 - tid;                 // (SD_CAND == SD_BLOCK: at most one candidate per thread)
        if (e < total) {
            nv = cand_v[e]; ncol = cand_c[e];
            int lo = 0, hi = cnt;                         // list entries that beat it: the list is sorted
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (sd_beats(top_v[mid], top_c[mid], nv, ncol)) lo = mid + 1; else hi = mid; }
            int b = lo;
            for (int x = 0; x < total; ++x) b += sd_beats(cand_v[x], cand_c[x], nv, ncol) ? 1 : 0;
            if (nv != -INFINITY) r_new = b;
        }
        sd_barrier();                                     // every read of the old list is done
        if (r_list < K) { top_v[r_list] = lv; top_c[r_list] = lc; }
        if (r_new < K) { top_v[r_new] = nv; top_c[r_new] = ncol; }
        if (tid == 0) { s_cnt = min(K, cnt + total - s_dead); s_nq = 0; s_dead = 0; }      // (everybody read them before the barrier above)
        sd_barrier();
    };
    // exact fallback: the K best of (list U the window's candidates) by K rounds of a block-wide argmax over the sums that
    // are still in the threads' own table slots; the slots are left clean.
    auto argmax_rounds = [&](int colA, double floor_v, bool strict) __attribute__((always_inline)) {
        const int cnt = s_cnt;
        uint64_t taken = 0;                               // bit k: slot k is out of the race; bit SD_SPT: my list entry
        double lv = 0.0; int lc = II_NONE;                // my list entry (threads below cnt)
        if (tid < cnt) { lv = top_v[tid]; lc = top_c[tid]; } else taken |= 1ull << SD_SPT;
        for (int k = 0; k < SD_SPT; ++k) {
            const double vk = tab[k * SD_BLOCK + tid];
            if (!(vk != 0.0 && (strict ? vk > floor_v : vk >= floor_v))) taken |= 1ull << k;      // (the candidates: all that is left)
        }
        sd_barrier();
        int filled = 0;
        for (int r = 0; r < K; ++r) {
            double bv = -INFINITY; int bc = II_NONE;
            for (int k = 0; k < SD_SPT; ++k) {
                const int col = colA + k * SD_BLOCK + tid;
                const double vk = tab[k * SD_BLOCK + tid];
                if (!(taken & (1ull << k)) && (bc == II_NONE || sd_beats(vk, col, bv, bc))) { bv = vk; bc = col; }
            }
            if (!(taken & (1ull << SD_SPT)) && (bc == II_NONE || sd_beats(lv, lc, bv, bc))) { bv = lv; bc = lc; }
            double wv = bv; int wc = bc;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const double ov = __shfl_xor(wv, off);
                const int oc = __shfl_xor(wc, off);
                if (oc != II_NONE && (wc == II_NONE || sd_beats(ov, oc, wv, wc))) { wv = ov; wc = oc; }
            }
            if (lane == 0) { wred_v[wave] = wv; wred_c[wave] = wc; }
            sd_barrier();
            double gv = wred_v[0]; int gc = wred_c[0];
            for (int w = 1; w < SD_BLOCK / 64; ++w)
                if (wred_c[w] != II_NONE && (gc == II_NONE || sd_beats(wred_v[w], wred_c[w], gv, gc))) { gv = wred_v[w]; gc = wred_c[w]; }
            sd_barrier();
            if (gc == II_NONE) break;
            if (tid == 0) { top_v[r] = gv; top_c[r] = gc; }
            ++filled;
            if (gc == bc) {                               // mine: retire it
                if (!(taken & (1ull << SD_SPT)) && lc == gc) taken |= 1ull << SD_SPT;
                else {
                    for (int k = 0; k < SD_SPT; ++k) if (colA + k * SD_BLOCK + tid == gc) taken |= 1ull << k;
                }
            }
        }
        for (int k = 0; k < SD_SPT; ++k) tab[k * SD_BLOCK + tid] = 0.0;
        sd_barrier();
        if (tid == 0) { s_cnt = filled; s_nq = 0; }
        sd_barrier();
    };

    // ---- after the window's entries are in the table: exclusions, scan, candidates.  The scan keeps NOTHING in registers
    // across the window: a thread reads its 16 sums for a running maximum, ONE compare against the list's K-th sum says
    // whether it holds a candidate at all (after the first windows almost never), and it clears its slots; only a thread
    // with candidates reads its slots again.  (Swapping the slots for zero into 16 register pairs cost 32 VGPRs, which
    // evicted the prefetched entries of the next window: the compiler drained every load in flight at that point.)
    auto scan_and_select = [&](int colA) __attribute__((always_inline)) {
        poison(colA, colA + SP_CHUNK);
        sd_barrier();
        const int cnt = s_cnt;
        double floor_v;                                   // candidates: sums strictly above (list full) / at or above (short list)
        bool strict;
        if (cnt == K) {
            floor_v = top_v[K - 1]; strict = true;
        } else {
            // short list: the wave's K-th best thread maximum is a lower bound of the window's K-th best sum
            double lm = -INFINITY;
            for (int k = 0; k < SD_SPT; ++k) { const double vk = tab[k * SD_BLOCK + tid]; lm = (vk != 0.0 && vk > SD_LIVE) ? fmax(lm, vk) : lm; }
            double vk = -INFINITY;
            for (int r = 0; r < Kw; ++r) {
                vk = sd_wave_max(lm);
                if (vk == -INFINITY) break;
                const unsigned long long b = __ballot(lm == vk);
                if (lane == __ffsll((long long)b) - 1) lm = -INFINITY;
            }
            if (lane == 0) wred_v[wave] = vk;
            sd_barrier();
            floor_v = wred_v[0];
#pragma unroll
            for (int w = 1; w < SD_BLOCK / 64; ++w) floor_v = split ? fmin(floor_v, wred_v[w]) : fmax(floor_v, wred_v[w]);
            floor_v = fmax(floor_v, SD_LIVE);
            strict = false;
        }
        double mx = tab[tid];
#pragma unroll
        for (int k = 1; k < SD_SPT; ++k) mx = fmax(mx, tab[k * SD_BLOCK + tid]);
        uint32_t mine = 0;                                // my slots that hold a candidate: cleared after the ranking
        if (strict ? mx > floor_v : mx >= floor_v) {
            for (int k = 0; k < SD_SPT; ++k) {
                const double vk = tab[k * SD_BLOCK + tid];
                if (vk != 0.0 && (strict ? vk > floor_v : vk >= floor_v)) {
                    mine |= 1u << k;
                    const int pos = atomicAdd(&s_nq, 1);
                    if (pos < cap) { cand_v[pos] = vk; cand_c[pos] = colA + k * SD_BLOCK + tid; }
                }
            }
        }
        if (__any(mine != 0u)) {
#pragma unroll
            for (int k = 0; k < SD_SPT; ++k) if (!(mine & (1u << k))) tab[k * SD_BLOCK + tid] = 0.0;
        } else {                                          // the common case, branch-free: nobody in the wave holds a candidate
#pragma unroll
            for (int k = 0; k < SD_SPT; ++k) tab[k * SD_BLOCK + tid] = 0.0;
        }
        sd_barrier();                                  // appends done; every slot without a candidate is clean
        const int total = s_nq;
        if (total > cap) argmax_rounds(colA, floor_v, strict);      // more candidates than the buffer holds (mass ties): exact rounds
        else if (total > 0) {
            for (int k = 0; k < SD_SPT; ++k) if (mine & (1u << k)) tab[k * SD_BLOCK + tid] = 0.0;
            rank_merge(false);
        }
    };

    // ---------------------------------------------------------------- main sweep
    if (single) {
        // The common case (history <= SD_HIST rows): a thread keeps ONE row for the whole sweep, and the loop is software
        // pipelined -- the entries of window c + 1 (SD_PRE per lane) and the row's entry position two windows ahead are
        // requested at the top of window c, so that no global round trip sits between two barriers (without this the
        // kernel spends ~4 us per window waiting on two dependent loads: 125 ms per 1 M users at C5).
        const int L = max(1, SD_BLOCK / max(m1, 1));      // lanes per history row (any value, not only powers of two)
        const int e = tid / L, l = tid - e * L;
        const bool has_row = e < m1;
        const int64_t *cpb = p.cptr + (int64_t)(has_row ? hj[e] : 0) * nc1;
        const double x = has_row ? ldexp((double)xf[e], p.fx_shift) : 0.0;      // history weight * 2^shift
        // entry positions as 32-bit offsets from the row's first entry (a row has fewer than 2^31 entries); IDX32: the whole
        // matrix has fewer than 2^30 entries, so a 32-bit index addresses any of them from the (uniform) array base
        const int64_t row0q = has_row ? cpb[0] : 0;
        // the entries as interleaved (column, value) pairs (m_pair, built at set time): ONE 8-byte load per entry, and a row's
        // segment of a window -- ~8 entries -- is 64 contiguous bytes.  With separate index / value arrays each segment
        // touched two cache lines for 32 useful bytes each, and the L2 -> L1 line traffic (4 x the useful bytes per array,
        // fetched again by the next window) set the pace: 105 ms per 1 M users at C5 whatever the instruction count.
        const uint2 *const rp = p.m_pair + (IDX32 ? 0 : row0q);
        const uint32_t base32 = IDX32 ? (uint32_t)row0q : 0u;
        struct Pre { uint2 ent[SD_PRE]; int first, end; };
        // Every load below is UNCONDITIONAL (out-of-range lanes read a clamped, valid address and are masked at use) and
        // nothing computed from a loaded value is touched before the next window: the compiler can then count the loads
        // in flight (s_waitcnt vmcnt(n)) instead of draining them all, which is what makes the prefetch overlap at all.
        auto issue = [&](Pre &s, int qa, int qb) __attribute__((always_inline)) {
            const int last = max(qb - 1, 0);
#pragma unroll
            for (int t = 0; t < SD_PRE; ++t) s.ent[t] = rp[base32 + (uint32_t)min(qa + l + t * L, last)];
            s.first = qa + l;
            s.end = qb;
        };
        // branch-free: an entry past the segment adds 0.0 to one of the lane's own slots
        auto consume = [&](const Pre &s, int colA) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < SD_PRE; ++t) {
                const bool ok = s.first + t * L < s.end;
                const double r = fma(x, (double)__uint_as_float(s.ent[t].y), SD_MAGIC) - SD_MAGIC;
                atomicAdd(&tab[ok ? (int)s.ent[t].x - colA : SP_CHUNK + lane], ok ? r : 0.0);
            }
            // segments longer than the prefetch (histories beyond 85 rows leave a row one or two lanes: 3 or 6 of its ~8
            // entries per window): three loads in flight per round trip, not one
            for (int q = s.first + SD_PRE * L; q < s.end; q += SD_PRE * L) {
                const int last = s.end - 1;
                uint2 e2[SD_PRE];
#pragma unroll
                for (int t = 0; t < SD_PRE; ++t) e2[t] = rp[base32 + (uint32_t)min(q + t * L, last)];
#pragma unroll
                for (int t = 0; t < SD_PRE; ++t)
                    if (q + t * L < s.end) atomicAdd(&tab[(int)e2[t].x - colA], fma(x, (double)__uint_as_float(e2[t].y), SD_MAGIC) - SD_MAGIC);
            }
        };
        // (only the LOW 32 bits of an entry position are loaded: the difference to the row's first entry fits, and a 64-bit
        // load whose upper half is dead lets the register allocator reuse that half while the load is in flight -- which
        // costs a full drain of the loads at the next write to it)
        auto pos_lo = [&](int k) __attribute__((always_inline)) { return reinterpret_cast<const uint32_t *>(cpb + k)[0]; };
        const uint32_t row0lo = (uint32_t)row0q;
        int b1 = has_row ? (int)(pos_lo(1) - row0lo) : 0;                         // my row's offset at the start of window c + 1
        uint32_t b2raw = pos_lo(min(2, p.n_chunks));                              // ... of window c + 2, as loaded (converted when used)
        Pre A, B;
        issue(A, 0, b1);
        SD_STAMP(9);                                      // 9: prologue
        // ---- the window's selection, ENTRY-driven: after the barrier every lane swaps the slot of each of its entries for
        // zero.  The first lane to arrive at a column takes its (complete) sum, the others get 0: every touched slot is
        // read once, cleared, and owned by exactly one lane -- no scan of the 4096 slots, no poison for the exclusions (a
        // candidate is looked up in the user's exclusion row instead: after the first windows candidates are rare).
        // Lanes without an entry swap a private dummy slot behind the table.
        int cnt_r = 0, nq_prev = 0;                       // register copies of s_cnt / s_nq (candidates pending) at the window's start; the list's K-th sum
        double tau = 0.0;
        auto excluded = [&](int col) __attribute__((always_inline)) -> bool {
            if (n_ex == 0) return false;
            const int64_t w = ex_lower_bound(col);
            return w < n_ex && (ex_lds ? exl[w] : p.ex_idx[eb + w]) == col;
        };
        auto append = [&](double v, int col) __attribute__((always_inline)) {
            const int pos = atomicAdd(&s_nq, 1);
            if (pos < cap) { cand_v[pos] = v; cand_c[pos] = col; }
        };
        auto claim = [&](int slot) __attribute__((always_inline)) -> double {
            return __longlong_as_double((long long)atomicExch(reinterpret_cast<unsigned long long *>(&tab[slot]), 0ull));
        };
        auto select_claimed = [&](const Pre &s, int c) __attribute__((always_inline)) {
            const int colA = c * SP_CHUNK;
            const bool full = cnt_r == K;
            double v[SD_PRE];
#pragma unroll
            for (int t = 0; t < SD_PRE; ++t) {
                const bool ok = s.first + t * L < s.end;
                v[t] = claim(ok ? (int)s.ent[t].x - colA : SP_CHUNK + lane);
            }
            for (int q = s.first + SD_PRE * L; q < s.end; q += SD_PRE * L) {        // segments longer than the prefetch
                const int last = s.end - 1;
                int col[SD_PRE];
#pragma unroll
                for (int t = 0; t < SD_PRE; ++t) col[t] = (int)rp[base32 + (uint32_t)min(q + t * L, last)].x;
#pragma unroll
                for (int t = 0; t < SD_PRE; ++t) {
                    if (q + t * L >= s.end) break;
                    const double vv = claim(col[t] - colA);
                    if (vv != 0.0 && (!full || vv > tau)) append(vv, col[t]);      // (exclusions: looked up at the merge)
                }
            }
            if (full) {
                bool any = false;                         // (a zero passes when the K-th sum is negative: sorted out inside)
#pragma unroll
                for (int t = 0; t < SD_PRE; ++t) any |= v[t] > tau;
                if (any) {
#pragma unroll
                    for (int t = 0; t < SD_PRE; ++t)
                        if (v[t] != 0.0 && v[t] > tau) append(v[t], (int)s.ent[t].x);     // (exclusions: looked up at the merge)
                }
            } else {
                // short list: a lower bound of the window's K-th best sum.  Keys = the high words of the lanes' best sums,
                // order-preserving (0 = no sum); a bit-by-bit search (ballot + popcount, no cross-lane data movement)
                // finds the top SD_FLOOR_BITS bits of the wave's K-th largest key: at least K lanes -- K distinct columns,
                // a column's sum is held by one lane only -- are at or above it.  Excluded columns are looked up first
                // (they must not lift the bound).
                uint32_t kmax = 0, key[SD_PRE];
#pragma unroll
                for (int t = 0; t < SD_PRE; ++t) {
                    if (v[t] != 0.0 && excluded((int)s.ent[t].x)) v[t] = 0.0;
                    const uint32_t hi = (uint32_t)__double2hiint(v[t]);
                    key[t] = v[t] != 0.0 ? (hi ^ ((uint32_t)((int32_t)hi >> 31) | 0x80000000u)) : 0u;
                    kmax = max(kmax, key[t]);
                }
                uint32_t fl = 0;
#pragma unroll
                for (int b = 31; b > 31 - SD_FLOOR_BITS; --b) {
                    const uint32_t trial = fl | (1u << b);
                    if (__popcll(__ballot(kmax >= trial)) >= Kw) fl = trial;
                }
                if (lane == 0) wred_n[wave] = fl;
                sd_barrier();
                uint32_t floor_k = wred_n[0];
#pragma unroll
                for (int w = 1; w < SD_BLOCK / 64; ++w) floor_k = split ? min(floor_k, wred_n[w]) : max(floor_k, wred_n[w]);
                floor_k = max(floor_k, 1u);
#pragma unroll
                for (int t = 0; t < SD_PRE; ++t)
                    if (key[t] >= floor_k) append(v[t], (int)s.ent[t].x);
            }
            SD_STAMP(4);                                  // 4: claims + candidate appends
            sd_barrier();                                 // appends done, every touched slot is clean
            SD_STAMP(5);                                  // 5: second barrier
            const int total = s_nq;
            if (total > cap) {
                // more candidates than the buffer holds (mass ties): the table is clean, so -- after the candidates of the
                // earlier windows are merged -- the window is accumulated again and ranked by the scanning path (exact
                // argmax rounds)
                sd_barrier();                             // (everybody has read s_nq)
                if (tid == 0) s_nq = nq_prev;
                sd_barrier();
                if (nq_prev > 0) rank_merge(true);
                accumulate_rows(m1, c, colA);
                scan_and_select(colA);
                cnt_r = s_cnt; tau = top_v[K - 1]; nq_prev = 0;
            } else if (total > 0 && (!full || total >= lazy || total + lazy > cap)) {
                // LAZY merge: while the list is full, new candidates wait behind it (the stale K-th sum only admits a few
                // more) until SD_LAZY + K of them are pending: ~4 merges per user instead of one in every second window
                rank_merge(true);
                cnt_r = s_cnt; tau = top_v[K - 1]; nq_prev = 0;
            } else nq_prev = total;
        };
        auto window = [&](int c, const Pre &cur, Pre &nxt) __attribute__((always_inline)) {
            const int b2 = has_row ? (int)(b2raw - row0lo) : 0;     // requested a whole window ago
            issue(nxt, b1, b2);                                     // window c + 1 (an empty range past the last window)
            b2raw = pos_lo(min(c + 3, p.n_chunks));
            SD_STAMP(0);                                            // 0: issue (incl. the wait for the loads of the last window)
            consume(cur, c * SP_CHUNK);
            b1 = b2;
            SD_STAMP(1);                                            // 1: consume
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            SD_STAMP(2);                                            // 2: the adds have landed
            __builtin_amdgcn_s_barrier();                           // the window's sums are complete
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            SD_STAMP(3);                                            // 3: barrier skew
            select_claimed(cur, c);
            SD_STAMP(7);                                            // 7: decisions, merges
        };
        for (int c = 0; c < p.n_chunks; c += 2) {
            window(c, A, B);
            if (c + 1 < p.n_chunks) window(c + 1, B, A);
        }
        if (nq_prev > 0) rank_merge(true);                          // the candidates still pending
        SD_STAMP(7);
    } else {
        for (int c = 0; c < p.n_chunks; ++c) {
            accumulate(c, c * SP_CHUNK);
            scan_and_select(c * SP_CHUNK);
        }
    }

    // ---------------------------------------------------------------- zero sums (score 0 * col_scale): only when they can still rank
    {
        int found = 0;
        for (int c = 0; c < p.n_chunks && found < K; ++c) {
            const int colA = c * SP_CHUNK;
            const int cnt = s_cnt;
            const double tv = cnt == K ? top_v[K - 1] : 0.0;
            const int tc = cnt == K ? top_c[K - 1] : II_NONE;
            if (cnt == K && !sd_beats(0.0, colA, tv, tc)) break;      // not even the window's first column would rank
            sd_barrier();
            accumulate(c, colA);
            poison(colA, colA + SP_CHUNK);
            sd_barrier();
            // contiguous slots per thread here: zero columns are taken in COLUMN order
            int mine = 0;
            uint32_t okbits = 0;
#pragma unroll
            for (int k = 0; k < SD_SPT; ++k) {
                const int lc = SD_SPT * tid + k;
                const int col = colA + lc;
                bool ok = tab[lc] == 0.0 && col < p.n_items;
                if (ok && p.pen != nullptr) { const float z = 0.0f * p.pen[col]; ok = z == z; }
                if (ok) ok = cnt < K || sd_beats(0.0, col, tv, tc);
                if (ok) { okbits |= 1u << k; ++mine; }
            }
            int incl = mine;                                                  // block exclusive prefix of `mine`
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if (lane >= off) incl += t; }
            if (lane == 63) wred_n[wave] = (uint32_t)incl;
            sd_barrier();
            int before = incl - mine, all = 0;
            for (int w = 0; w < SD_BLOCK / 64; ++w) { if (w < wave) before += (int)wred_n[w]; all += (int)wred_n[w]; }
#pragma unroll
            for (int k = 0; k < SD_SPT; ++k) tab[SD_SPT * tid + k] = 0.0;    // my slots: clean for the next window
            before += found;
            // at most K zero columns rank in all (they tie: column order), so at most K - found enter here
            for (int k = 0; k < SD_SPT && before < K; ++k)
                if (okbits & (1u << k)) {
                    const int pos = atomicAdd(&s_nq, 1);
                    cand_v[pos] = 0.0; cand_c[pos] = colA + SD_SPT * tid + k;     // K - found <= K <= cap
                    ++before;
                }
            found += all;
            sd_barrier();
            if (s_nq > 0) rank_merge(false);
        }
    }
    sd_barrier();

    const int cnt = s_cnt;
    if (tid < K) {
        int32_t id = -1;
        float sc = -INFINITY;
        if (tid < cnt) {
            id = top_c[tid];
            sc = (float)ldexp(top_v[tid], -p.fx_shift);                       // the exact sum scaled back, ONE rounding to fp32
            if (sc == 0.0f) sc = 0.0f;                                        // (a zero sum is +0: 0 * col_scale was finite)
        }
        out_ids[i * K + tid] = id;
        out_scores[i * K + tid] = sc;
    }
#ifdef IFR_ABLATE
    SD_STAMP(8);                                                              // 8: zero sweep + output
    if (tid == 0) {
        for (int k = 0; k < 10; ++k) atomicAdd(&sd_stamps[k], sd_acc[k]);
        atomicAdd(&sd_stamps[15], 1ull);
    }
#endif
    if (LIST) sd_barrier();                                                   // (the shared state is reused)
    } while (LIST && (it += gridDim.x) < limit);
}

// ------------------------------------------------------------------------------------
// The same sweep with ONE WAVE per user (round 3, last step): the workgroup-per-user kernel above spends its time in a
// dependent chain per window -- adds, LDS drain, barrier, exchanges, barrier, counter read -- that only the four
// workgroups a CU holds can overlap, and all four waves of a workgroup execute the same scalar bookkeeping.  A user
// with at most 256 history rows needs neither: a wave keeps a row (longer histories: two, three or four) per lane, a
// private table of 1 024 direct-mapped fp64 accumulators (8 KB) and its candidate buffer; nothing crosses waves, so there is no s_barrier
// and no shared counter -- candidate slots come from a ballot prefix -- and a CU holds 16 independent users, each a
// workgroup of its own (with four users per workgroup the LDS of three finished waves waited for the slowest: K = 32
// 17 % slower).  Arithmetic, exchange-claim selection, lazy merges and the exact-domain ranking are those of the kernel
// above (same results bit for bit).
// What this kernel does not serve it hands to the workgroup kernel through p.long_list: histories beyond 512 rows (one-word layout),
// users whose list ends short or with a K-th sum <= 0 (zero-sum columns may rank: the sweep over them lives there),
// and a candidate buffer that overflows (mass ties).
// ------------------------------------------------------------------------------------
constexpr int SW_WIN = 1024;                            // columns per window
constexpr int SW_CAND = 128;                            // the list + pending candidates (two per lane in the merge)
constexpr int SW_ROWS = 4;                              // history rows per lane, at most (histories of up to 256 rows)
constexpr int SW_TAIL = 256;                            // history rows beyond four per lane that the one-word sweep still takes (two dependent loads per window each)
constexpr int SW_MAXK = 64;                             // longest list of this kernel (longer ones: the workgroup kernel)
constexpr int SW_EXL = 32;                              // exclusion row entries kept in LDS
constexpr int SW_WAVES = 1;                             // users (waves) per workgroup
// the per-row table of entry offsets per window: T[SW_CW_LEAD + k] = offset of the row's first entry with column >= k * SW_WIN,
// k clamped into [0, n_win] (so that offsets a few windows past the end read as empty segments without a clamp), rows
// 16-byte aligned; the kernel reads two offsets per load
constexpr int SW_CW_LEAD = 2, SW_CW_TAIL = 8;
__host__ __device__ inline int sw_cw_stride(int n_win) { return (SW_CW_LEAD + n_win + 1 + SW_CW_TAIL + 3) & ~3; }

struct SwLds {
    double tab[SW_WIN];
    double cv[SW_CAND];
    int cc[SW_CAND];
    int exl[SW_EXL];
};

__device__ __forceinline__ void sw_lds_order()
{
    // LDS operations of one wave are performed in order, but the compiler must not move them across this point, and the
    // returning ones must have landed: wait for the wave's outstanding LDS traffic
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// WORDS = window words per (row, window): 1, 2 or 3 (3, 6 or 9 entries), chosen when the matrix is set from its segment
// lengths; a lane keeps at most 4 / WORDS rows (registers).
template <int WORDS>
__global__ __launch_bounds__(64 * SW_WAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) void item_sparse_wave_kernel(ItemItemParams p, int32_t *out_ids, float *out_scores)
{
    constexpr int RMAX = WORDS == 1 ? SW_ROWS : (WORDS == 2 ? 3 : 2);      // rows per lane, at most (registers: 4, 6, 6 words per buffer)
    __shared__ __attribute__((aligned(16))) SwLds lds[SW_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    SwLds &S = lds[wave];
    const int64_t i = (int64_t)blockIdx.x * SW_WAVES + wave;
    if (i >= p.n) return;                                // (the whole wave)
    const int u = p.user_ids[i];
    const int K = p.K;
    const int n_win = p.n_win;
    const unsigned long long below = (1ull << lane) - 1ull;

    int64_t hb = 0, he = 0, eb = 0, ee = 0;
    if (u >= 0 && u < p.h_rows) { hb = p.h_ptr[u]; he = p.h_ptr[u + 1]; }
    if (p.ex_ptr != nullptr && u >= 0 && u < p.ex_rows) { eb = p.ex_ptr[u]; ee = p.ex_ptr[u + 1]; }
    const int64_t n_ex = ee - eb;
    const bool ex_lds = n_ex <= SW_EXL;
    auto hand_over = [&]() __attribute__((always_inline)) {
        if (lane == 0) p.long_list[atomicAdd(p.long_cnt, 1)] = (int32_t)i;
    };
    // (one-word layout: up to SW_TAIL more rows ride along without a lane of their own, see the sweep)
    if (he - hb > 64 * RMAX + (WORDS == 1 ? SW_TAIL : 0) || he == hb) { hand_over(); return; }      // (no history: only zero-sum columns to rank)
    const int m = (int)(he - hb);

    for (int w = lane; w < SW_WIN; w += 64) S.tab[w] = 0.0;
    if (ex_lds) for (int w = lane; w < (int)n_ex; w += 64) S.exl[w] = p.ex_idx[eb + w];
    auto excluded = [&](int col) __attribute__((always_inline)) -> bool {
        int64_t lo = 0, hi = n_ex;
        if (ex_lds) { while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (S.exl[mid] < col) lo = mid + 1; else hi = mid; } }
        else { while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (p.ex_idx[eb + mid] < col) lo = mid + 1; else hi = mid; } }
        return lo < n_ex && (ex_lds ? S.exl[lo] : p.ex_idx[eb + lo]) == col;
    };

    // A lane keeps ONE history row for the whole sweep -- or two / four rows (lane, lane + 64, ...: histories of up to 256
    // rows).  The row's entries come from the WINDOW WORDS (m_win, built at set time): one 16-byte word per (row, window)
    // that packs up to three entries -- three fp32 values + 3 x 10 bits of column inside the window + a flag --
    // so a window costs a lane ONE load at an address that is a plain stride (no entry offsets to fetch first).  What a
    // wave's load costs here is the cache-line look-up per lane, not the bytes: 8-byte entries took three loads + an
    // offset load per window, then a 16-byte + an 8-byte load + a quarter of an offset load (20.7 -> 24.4 M users/s at
    // C5), now one.  A segment with more than three entries is flagged in its word and its other entries are read from
    // the pair array.
    const uint2 *const rp = p.m_pair;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 *const mw = reinterpret_cast<const u32x4 *>(p.m_win);
    auto claim = [&](int slot) __attribute__((always_inline)) -> double {
        return __longlong_as_double((long long)atomicExch(reinterpret_cast<unsigned long long *>(&S.tab[slot]), 0ull));
    };

    // list = cv/cc[0, cnt) (sorted), pending candidates behind it up to nq: wave-uniform registers, no shared counter
    int cnt = 0, nq = 0;
    bool over = false;
    double tau = 0.0;
    auto append = [&](bool c, double v, int col) __attribute__((always_inline)) {
        const unsigned long long b = __ballot(c);
        if (b != 0ull) {
            const int pos = nq + __popcll(b & below);
            if (c && pos < SW_CAND) { S.cv[pos] = v; S.cc[pos] = col; }
            nq += __popcll(b);
            over |= nq > SW_CAND;
        }
    };
    // the list U the pending candidates -> the new list, by counting (in place: every rank is computed before any write)
    auto merge = [&]() __attribute__((always_inline)) {
        sw_lds_order();
        const int total = nq;
        const int e0 = lane, e1 = lane + 64;
        double v0 = -INFINITY, v1 = -INFINITY;
        int c0 = 0, c1 = 0;
        if (e0 < total) { v0 = S.cv[e0]; c0 = S.cc[e0]; }
        if (e1 < total) { v1 = S.cv[e1]; c1 = S.cc[e1]; }
        // the new candidates are looked up in the user's exclusion row only here
        if (n_ex != 0) {
            if (e0 >= cnt && e0 < total && excluded(c0)) { v0 = -INFINITY; S.cv[e0] = -INFINITY; }
            if (e1 >= cnt && e1 < total && excluded(c1)) { v1 = -INFINITY; S.cv[e1] = -INFINITY; }
        }
        const int dead = __popcll(__ballot(e0 >= cnt && e0 < total && v0 == -INFINITY)) + __popcll(__ballot(e1 >= cnt && e1 < total && v1 == -INFINITY));
        sw_lds_order();
        int r0 = 0, r1 = 0;
        for (int q = 0; q < total; ++q) {
            const double vq = S.cv[q];
            const int cq = S.cc[q];
            r0 += sd_beats(vq, cq, v0, c0) ? 1 : 0;
            r1 += sd_beats(vq, cq, v1, c1) ? 1 : 0;
        }
        sw_lds_order();
        if (e0 < total && r0 < K && v0 != -INFINITY) { S.cv[r0] = v0; S.cc[r0] = c0; }
        if (e1 < total && r1 < K && v1 != -INFINITY) { S.cv[r1] = v1; S.cc[r1] = c1; }
        sw_lds_order();
        cnt = min(K, total - dead);
        nq = cnt;
        tau = S.cv[K - 1];                                // (meaningful once cnt == K)
    };

    // compiled for one, two and four rows per lane
    auto sweep = [&](auto rows_t) __attribute__((always_inline)) {
    constexpr int ROWS = decltype(rows_t)::value;
    bool has_row[ROWS];
    double x[ROWS];
    int hj[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int er = lane + 64 * r;
        has_row[r] = er < m;
        hj[r] = has_row[r] ? p.h_idx[hb + er] : 0;
        x[r] = has_row[r] ? ldexp((double)p.h_val[hb + er], p.fx_shift) : 0.0;
    }
    struct Pre { u32x4 w[ROWS][WORDS]; };
    // (unconditional loads at valid addresses: a lane without a row reads row 0 and ignores it)
    auto issue = [&](Pre &s, int c) __attribute__((always_inline)) {
        const int cc = min(c, n_win - 1);
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int j = 0; j < WORDS; ++j) s.w[r][j] = mw[((uint32_t)hj[r] * (uint32_t)n_win + (uint32_t)cc) * WORDS + j];
    };
    // A word always holds three (column, value) pairs: where the segment has fewer entries the spare pairs carry the
    // value 0 and the column of a real pair (or column 0).  Adding 0 changes nothing, and the exchange that follows
    // may well return the slot's sum to this lane instead of the lane whose entry it is -- whoever claims a slot reports
    // its column, which the slot number gives.  So the common window has no counts and no per-entry masks.
    // (pair t of a row's segment: pair t % 3 of word t / 3)
    auto col_of = [&](const Pre &s, int r, int t) __attribute__((always_inline)) -> int { return (int)((s.w[r][t / 3][3] >> (10 * (t % 3))) & 1023u); };
    auto val_of = [&](const Pre &s, int r, int t) __attribute__((always_inline)) -> float { return __uint_as_float(s.w[r][t / 3][t % 3]); };
    // (a lane without a row keeps to slot `lane` of the table -- forty lanes adding zeros to row 0's slots would queue up
    // at the same LDS address -- and, like any lane, reports what an exchange there hands it under the SLOT's column)
    auto slot_of = [&](const Pre &s, int r, int t) __attribute__((always_inline)) -> int { return has_row[r] ? col_of(s, r, t) : lane; };
    auto more_of = [&](const Pre &s, int r) __attribute__((always_inline)) -> bool { return has_row[r] && (s.w[r][0][3] >> 30) != 0u; };
    // entries 3 ... of a marked segment: [q0, q1) in the pair array
    auto rest_of_row = [&](int row, int c, uint32_t &q0, uint32_t &q1) __attribute__((always_inline)) {
        const uint32_t *const cw = p.cptr_w + (int64_t)row * sw_cw_stride(n_win) + SW_CW_LEAD;
        const uint32_t base = (uint32_t)p.cptr[(int64_t)row * (p.n_chunks + 1)];        // the row's first entry (< 2^30 entries in all)
        q0 = base + cw[c] + 3u * WORDS;
        q1 = base + cw[c + 1];
    };
    auto rest_of = [&](int r, int c, uint32_t &q0, uint32_t &q1) __attribute__((always_inline)) { rest_of_row(hj[r], c, q0, q1); };
    // Histories a little beyond four rows per lane (one-word layout only; 0.4 % of C5's users, who cost the workgroup
    // kernel 7 % of the call): the rows past 256 have no lane and no prefetch -- their words are fetched inside the
    // window, for the adds and again for the exchanges -- and such a window always takes the slow path.
    constexpr bool TAIL = WORDS == 1 && ROWS == RMAX;
    const int tail0 = 64 * ROWS;
    const bool has_tail = TAIL && m > tail0;
    auto consume = [&](const Pre &s, int c, int colA) __attribute__((always_inline)) {
        bool more = false;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
#pragma unroll
            for (int t = 0; t < 3 * WORDS; ++t)
                atomicAdd(&S.tab[slot_of(s, r, t)], fma(x[r], (double)val_of(s, r, t), SD_MAGIC) - SD_MAGIC);
            more |= more_of(s, r);
        }
        if (TAIL && has_tail) {
            for (int er = tail0 + lane; er < m; er += 64) {
                const int row = p.h_idx[hb + er];
                const double xt = ldexp((double)p.h_val[hb + er], p.fx_shift);
                const u32x4 w = mw[(uint32_t)row * (uint32_t)n_win + (uint32_t)c];
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    atomicAdd(&S.tab[(w[3] >> (10 * t)) & 1023u], fma(xt, (double)__uint_as_float(w[t]), SD_MAGIC) - SD_MAGIC);
                if ((w[3] >> 30) != 0u) {
                    uint32_t q0, q1;
                    rest_of_row(row, c, q0, q1);
                    for (uint32_t q = q0; q < q1; ++q) {
                        const uint2 e2 = rp[q];
                        atomicAdd(&S.tab[(int)e2.x - colA], fma(xt, (double)__uint_as_float(e2.y), SD_MAGIC) - SD_MAGIC);
                    }
                }
            }
        }
        if (__builtin_expect(__any(more) != 0, 0)) {      // segments with more than three entries (rare on the matrices this kernel is picked for)
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
                if (more_of(s, r)) {
                    uint32_t q0, q1;
                    rest_of(r, c, q0, q1);
                    for (uint32_t q = q0; q < q1; ++q) {
                        const uint2 e2 = rp[q];
                        atomicAdd(&S.tab[(int)e2.x - colA], fma(x[r], (double)__uint_as_float(e2.y), SD_MAGIC) - SD_MAGIC);
                    }
                }
        }
    };
    Pre A, B;
    issue(A, 0);
    // The common window -- the list is full and no claimed sum beats its K-th -- is kept to the load, the adds, the
    // exchanges and ONE vote; everything else (candidates, marked segments, the short list of the first window, merges)
    // sits behind that vote.
    double tau_eff = -INFINITY;                           // the K-th sum while the list is full, else -inf (every sum is a candidate)
    auto window = [&](int c, const Pre &cur, Pre &nxt) __attribute__((always_inline)) {
        const int colA = c * SW_WIN;
        issue(nxt, c + 1);                                // (past the last window: the last one again, never used)
        consume(cur, c, colA);
        // the window's sums are complete: the LDS performs one wave's operations in issue order, so the exchanges below
        // find every lane's adds without waiting for them (only the compiler must keep the order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        double v[ROWS][3 * WORDS];
        bool look = false;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
#pragma unroll
            for (int t = 0; t < 3 * WORDS; ++t) v[r][t] = claim(slot_of(cur, r, t));
            look |= more_of(cur, r);
        }
        unsigned long long any = __ballot(look);          // (compares, not a maximum tree: fmax canonicalises its operands, an instruction each)
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int t = 0; t < 3 * WORDS; ++t) any |= __ballot(v[r][t] > tau_eff);
        if (TAIL && has_tail) any = ~0ull;
        if (__builtin_expect(any == 0ull, 1)) return;
        const bool full = cnt == K;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const bool mr = more_of(cur, r);
            if (__any(mr)) {
                uint32_t q0 = 0, q1 = 0;
                if (mr) rest_of(r, c, q0, q1);
                for (uint32_t q = q0; __any(q < q1); ++q) {      // (wave-uniform trip count for the ballots)
                    const bool in = q < q1;
                    const int col = in ? (int)rp[q].x : 0;
                    const double vv = in ? claim(col - colA) : 0.0;
                    append(in && vv != 0.0 && (!full || vv > tau), vv, col);
                }
            }
        }
        if (full) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
#pragma unroll
                for (int t = 0; t < 3 * WORDS; ++t) append(v[r][t] != 0.0 && v[r][t] > tau, v[r][t], colA + slot_of(cur, r, t));   // (a zero passes the vote when the K-th sum is negative)
        } else if ([&]() { bool nz = false;
#pragma unroll
                           for (int r = 0; r < ROWS; ++r)
#pragma unroll
                               for (int t = 0; t < 3 * WORDS; ++t) nz |= v[r][t] != 0.0;
                           return __any(nz) != 0; }()) {
            // short list: a lower bound of the window's K-th best sum by a bit-by-bit search over the order-preserving
            // high words of ALL claimed sums (ballot + popcount; excluded columns are looked up first)
            uint32_t key[ROWS][3 * WORDS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
#pragma unroll
                for (int t = 0; t < 3 * WORDS; ++t) {
                    if (v[r][t] != 0.0 && n_ex != 0 && excluded(colA + slot_of(cur, r, t))) v[r][t] = 0.0;
                    const uint32_t hi = (uint32_t)__double2hiint(v[r][t]);
                    key[r][t] = v[r][t] != 0.0 ? (hi ^ ((uint32_t)((int32_t)hi >> 31) | 0x80000000u)) : 0u;
                }
            uint32_t fl = 0;
#pragma unroll
            for (int b = 31; b > 31 - SD_FLOOR_BITS; --b) {
                const uint32_t trial = fl | (1u << b);
                int n_at = 0;
#pragma unroll
                for (int r = 0; r < ROWS; ++r)
#pragma unroll
                    for (int t = 0; t < 3 * WORDS; ++t) n_at += __popcll(__ballot(key[r][t] >= trial));
                if (n_at >= K) fl = trial;
            }
            const uint32_t floor_k = max(fl, 1u);
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
#pragma unroll
                for (int t = 0; t < 3 * WORDS; ++t) append(key[r][t] >= floor_k, v[r][t], colA + slot_of(cur, r, t));
        }
        if (TAIL && has_tail && !over) {
            // the rows without a lane: once the lanes' own rows have filled the list (else the workgroup kernel takes the user)
            if (nq > cnt && cnt < K) merge();
            if (cnt < K) over = true;
            for (int er0 = tail0; er0 < m && !over; er0 += 64) {
                const int er = er0 + lane;
                const bool in = er < m;
                const int row = in ? p.h_idx[hb + er] : 0;
                const u32x4 w = mw[(uint32_t)row * (uint32_t)n_win + (uint32_t)c];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int col = (int)((w[3] >> (10 * t)) & 1023u);
                    const double vv = in ? claim(col) : 0.0;
                    append(vv != 0.0 && vv > tau, vv, colA + col);
                }
                const bool mr = in && (w[3] >> 30) != 0u;
                if (__any(mr)) {
                    uint32_t q0 = 0, q1 = 0;
                    if (mr) rest_of_row(row, c, q0, q1);
                    for (uint32_t q = q0; __any(q < q1); ++q) {
                        const bool inq = q < q1;
                        const int col = inq ? (int)rp[q].x : 0;
                        const double vv = inq ? claim(col - colA) : 0.0;
                        append(vv != 0.0 && vv > tau, vv, col);
                    }
                }
                if (!over && nq + 24 + 3 * 64 > SW_CAND && nq > cnt) merge();      // (room for the next 64 rows' candidates)
            }
        }
        if (!over && nq > cnt && (cnt < K || nq - cnt >= SD_LAZY + K || nq + 24 > SW_CAND)) merge();   // lazy, like the kernel above
        tau_eff = cnt == K ? tau : -INFINITY;
    };
    for (int c = 0; c < n_win && !over; c += 2) {
        window(c, A, B);
        if (c + 1 < n_win && !over) window(c + 1, B, A);
    }
    };
    if constexpr (RMAX >= 4) {
        if (m > 192) sweep(std::integral_constant<int, 4>{});
        else if (m > 128) sweep(std::integral_constant<int, 3>{});
        else if (m > 64) sweep(std::integral_constant<int, 2>{});
        else sweep(std::integral_constant<int, 1>{});
    } else if constexpr (RMAX >= 3) {
        if (m > 128) sweep(std::integral_constant<int, 3>{});
        else if (m > 64) sweep(std::integral_constant<int, 2>{});
        else sweep(std::integral_constant<int, 1>{});
    } else if constexpr (RMAX >= 2) {
        if (m > 64) sweep(std::integral_constant<int, 2>{});
        else sweep(std::integral_constant<int, 1>{});
    } else sweep(std::integral_constant<int, 1>{});
    if (!over && nq > cnt) merge();
    // zero-sum columns (score 0 * col_scale) can still rank when the list is short or ends at or below zero: the sweep
    // over them lives in the workgroup kernel
    if (over || cnt < K || !(tau > 0.0)) { hand_over(); return; }
    if (lane < K) {
        out_ids[i * K + lane] = S.cc[lane];
        out_scores[i * K + lane] = (float)ldexp(S.cv[lane], -p.fx_shift);
    }
}

// entry offsets (from the row's first entry) of every row per SW_WIN-column window: [n_rows][n_win + 1]
__global__ void build_win_ptr_kernel(const int64_t *cptr, int nc1, const int32_t *idx, int64_t n_rows, int n_win, uint32_t *cw)
{
    const int stride = sw_cw_stride(n_win);
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows * stride) return;
    const int64_t r = t / stride;
    const int c = min(max((int)(t - r * stride) - SW_CW_LEAD, 0), n_win);
    const int64_t b = cptr[r * nc1], e = cptr[r * nc1 + nc1 - 1];
    int64_t lo = b, hi = e;
    const int64_t target = (int64_t)c * SW_WIN;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (idx[mid] < target) lo = mid + 1; else hi = mid; }
    cw[t] = (uint32_t)(lo - b);
}

// The window words of the wave-per-user kernel: per (row, window) one 16-byte word = the segment's first three entries
// (fp32 values in words 0 ... 2; word 3 = three 10-bit columns inside the window; bit 30 = the segment has more entries:
// the kernel fetches the rest from the pair array).  A segment with fewer than three entries fills the spare pairs with
// the value 0 and the column of its first entry (an empty one: a column that depends on the row): the kernel treats all
// three alike.
__global__ void build_win_words_kernel(const uint32_t *cw, const int64_t *cptr, int nc1, const uint2 *pair, int64_t n_rows, int n_win, int words, uint4 *out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows * n_win) return;
    const int64_t r = t / n_win;
    const int c = (int)(t - r * n_win);
    const uint32_t *row = cw + r * sw_cw_stride(n_win) + SW_CW_LEAD;
    const uint32_t q0 = row[c], q1 = row[c + 1];
    const int64_t base = cptr[r * nc1];
    const int n = (int)(q1 - q0);
    // the column the spare pairs point at: the segment's first entry (an empty segment: some column of the window, not
    // the same one for every row)
    const uint32_t spare = n > 0 ? pair[base + q0].x - (uint32_t)c * SW_WIN : (uint32_t)(r & (SW_WIN - 1));
    for (int j = 0; j < words; ++j) {
        uint2 e[3];
        for (int k = 0; k < 3; ++k) {
            const int q = 3 * j + k;
            if (q < n) { e[k] = pair[base + q0 + q]; e[k].x -= (uint32_t)c * SW_WIN; }
            else e[k] = make_uint2(spare, 0u);                        // (value +0.0f)
        }
        out[t * words + j] = make_uint4(e[0].y, e[1].y, e[2].y, e[0].x | (e[1].x << 10) | (e[2].x << 20) | (j == 0 && n > 3 * words ? 1u << 30 : 0u));
    }
}

int launch_build_win_words(const uint32_t *cw, const int64_t *cptr, int n_chunks, const uint2 *pair, int64_t n_rows, int n_win, int words, void *out, hipStream_t s)
{
    const int64_t total = n_rows * n_win;
    if (total <= 0) return 0;
    if (words < 1 || words > 3 || total * words >= ((int64_t)1 << 32)) return -1;
    hipLaunchKernelGGL(build_win_words_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cw, cptr, n_chunks + 1, pair, n_rows, n_win, words, (uint4 *)out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// (row, window) segments with more entries than a lane prefetches: the wave-per-user kernel walks those one load at a
// time, so it only pays on matrices whose rows spread evenly over the windows
// count[k]: segments with more than 3 (k + 1) entries, k = 0, 1, 2 (what one, two, three window words hold)
__global__ void count_long_segments_kernel(const uint32_t *cw, int64_t n_rows, int n_win, unsigned long long *count)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t len = 0;
    if (t < n_rows * n_win) {
        const int64_t r = t / n_win;
        const int c = (int)(t - r * n_win);
        const uint32_t *row = cw + r * sw_cw_stride(n_win) + SW_CW_LEAD;
        len = row[c + 1] - row[c];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const unsigned long long b = __ballot(len > 3u * (uint32_t)(k + 1));
        if ((threadIdx.x & 63) == 0 && b != 0ull) atomicAdd(count + k, (unsigned long long)__popcll(b));
    }
}

int launch_count_long_segments(const uint32_t *cw, int64_t n_rows, int n_win, unsigned long long *count, hipStream_t s)
{
    const int64_t total = n_rows * n_win;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(count_long_segments_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cw, n_rows, n_win, count);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int item_sparse_wave_windows(int n_items) { return (n_items + SW_WIN - 1) / SW_WIN; }
int item_sparse_wave_stride(int n_win) { return sw_cw_stride(n_win); }

int launch_build_win_ptr(const int64_t *cptr, int n_chunks, const int32_t *idx, int64_t n_rows, int n_win, uint32_t *cw, hipStream_t s)
{
    const int64_t total = n_rows * sw_cw_stride(n_win);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(build_win_ptr_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cptr, n_chunks + 1, idx, n_rows, n_win, cw);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int item_sparse_read_phases(unsigned long long *out16)
{
#ifdef IFR_ABLATE
    unsigned long long zero[16] = {0};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(sd_stamps), sizeof(zero)) != hipSuccess) return -3;
    if (hipMemcpyToSymbol(HIP_SYMBOL(sd_stamps), zero, sizeof(zero)) != hipSuccess) return -3;
    return 0;
#else
    (void)out16;
    return -1;
#endif
}

int launch_item_sparse_topk(const ItemItemParams &p, int32_t *out_ids, float *out_scores, hipStream_t stream)
{
    if (p.n <= 0 || p.n > (int64_t)0xffffffffu / SD_BLOCK || p.K < 1 || p.K > SD_MAXK) return -1;
    if (p.n_chunks < 1 || p.sp_cap < 0) return -1;
    // candidate buffer entries in use: all of them unless a test shrank it (never below K: the zero-sum sweep parks up to K)
    const int cap = p.sp_cap == 0 ? SD_CAND : (p.sp_cap < p.K ? p.K : (p.sp_cap > SD_CAND ? SD_CAND : p.sp_cap));
    // fewer than 2^30 entries (+ the over-read pad): 32-bit entry indices from the array bases (fewer address instructions)
    const bool idx32 = p.m_nnz + II_CSR_PAD < ((int64_t)1 << 30);
    if (p.cptr_w != nullptr && p.m_win != nullptr && p.long_list != nullptr && idx32 && p.K <= SW_MAXK) {
        // one wave per user; what that kernel hands over (long histories, lists that zero-sum columns may enter, mass
        // ties) is ranked by the workgroup kernel in list mode -- a fixed-size grid that reads the count on the device
        if (hipMemsetAsync(p.long_cnt, 0, sizeof(int), stream) != hipSuccess) return -3;
        ItemItemParams w = p;
        w.sp_cap = cap;
        const dim3 wgrid((unsigned)((p.n + SW_WAVES - 1) / SW_WAVES)), wblock(64 * SW_WAVES);
        if (p.win_words == 1) hipLaunchKernelGGL((item_sparse_wave_kernel<1>), wgrid, wblock, 0, stream, w, out_ids, out_scores);
        else if (p.win_words == 2) hipLaunchKernelGGL((item_sparse_wave_kernel<2>), wgrid, wblock, 0, stream, w, out_ids, out_scores);
        else if (p.win_words == 3) hipLaunchKernelGGL((item_sparse_wave_kernel<3>), wgrid, wblock, 0, stream, w, out_ids, out_scores);
        else return -1;
        const unsigned grid = (unsigned)(p.n < 2048 ? p.n : 2048);
        hipLaunchKernelGGL((item_sparse_topk_kernel<true, true>), dim3(grid), dim3(SD_BLOCK), 0, stream, w, out_ids, out_scores);
        return hipGetLastError() == hipSuccess ? 0 : -3;
    }
    ItemItemParams q = p;
    q.long_list = nullptr; q.long_cnt = nullptr;             // direct mode: workgroup b ranks user b
    q.sp_cap = cap;
    if (idx32)
        hipLaunchKernelGGL((item_sparse_topk_kernel<true, false>), dim3((unsigned)p.n), dim3(SD_BLOCK), 0, stream, q, out_ids, out_scores);
    else
        hipLaunchKernelGGL((item_sparse_topk_kernel<false, false>), dim3((unsigned)p.n), dim3(SD_BLOCK), 0, stream, q, out_ids, out_scores);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int item_sparse_pass_capacity() { return SD_CAND; }
int item_sparse_min_capacity() { return 1; }
int item_sparse_max_topk() { return SD_MAXK; }
int item_sparse_max_chunks() { return 1 << 20; }

// fp64 option: one wave per user ranks the n_lists * KL partial entries by counting (score descending on the DOUBLES,
// item index ascending) and writes the first K; the scores leave as fp32 (rounded once, after the ranking).
__global__ __launch_bounds__(64) void merge_topk64_kernel(const double *part_scores, const int32_t *part_ids, int n_lists, int KL, int K,
                                                          int32_t *out_ids, float *out_scores)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char m64_lds[];
    const int E = n_lists * KL;
    double *const es = reinterpret_cast<double *>(m64_lds);
    int *const ei = reinterpret_cast<int *>(es + E);
    const int64_t u = blockIdx.x;
    const int lane = threadIdx.x;
    for (int e = lane; e < E; e += 64) { es[e] = part_scores[u * E + e]; ei[e] = part_ids[u * E + e]; }
    __syncthreads();
    int n_valid = 0;
    for (int e = lane; e < E; e += 64) {
        const int id = ei[e];
        if (id == II_NONE) continue;
        const double s = es[e];
        ++n_valid;
        int rank = 0;
        for (int x = 0; x < E; ++x) {
            const int id2 = ei[x];
            const double s2 = es[x];
            rank += (id2 != II_NONE && ((s2 > s) || (s2 == s && id2 < id))) ? 1 : 0;
        }
        if (rank < K) { out_ids[u * K + rank] = id; out_scores[u * K + rank] = (float)s; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) n_valid += __shfl_xor(n_valid, off);
    for (int r = n_valid + lane; r < K; r += 64) { out_ids[u * K + r] = -1; out_scores[u * K + r] = -INFINITY; }
}

int launch_merge_topk_f64(const double *part_scores, const int32_t *part_ids, int64_t n, int n_lists, int KL, int K, int32_t *out_ids,
                          float *out_scores, hipStream_t stream)
{
    const size_t smem = (size_t)n_lists * KL * 12;
    if (n <= 0 || smem > 60000) return -1;
    hipLaunchKernelGGL(merge_topk64_kernel, dim3((unsigned)n), dim3(64), smem, stream, part_scores, part_ids, n_lists, KL, K, out_ids,
                       out_scores);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_item_item(const ItemItemParams &p, bool sparse, bool topk, hipStream_t stream)
{
    // HIP limits grid.x * block.x to 32 bits and grid.y to 65535
    if (p.n <= 0 || p.n_chunks <= 0 || p.n > (int64_t)0xffffffffu / II_BLOCK || p.n_chunks > 65535) return -1;
    const dim3 g((unsigned)p.n, (unsigned)p.n_chunks), b(II_BLOCK);
    if (sparse && p.out64 != nullptr) {                  // fp64 rows of a CSR matrix
        if (topk) return -1;
        hipLaunchKernelGGL((item_item_kernel<II_SPARSE_PASSES, true, false>), g, b, 0, stream, p);
        return hipGetLastError() == hipSuccess ? 0 : -3;
    }
    if (p.M64 != nullptr) {
        if (sparse) return -1;
        if (topk) hipLaunchKernelGGL((item_item_kernel<II_F64_PASSES, false, true, true>), g, b, 0, stream, p);
        else hipLaunchKernelGGL((item_item_kernel<II_F64_PASSES, false, false, true>), g, b, 0, stream, p);
        return hipGetLastError() == hipSuccess ? 0 : -3;
    }
    if (sparse) {
        if (topk) return -1;                             // CSR top-K is item_sparse_topk_kernel
        hipLaunchKernelGGL((item_item_kernel<II_SPARSE_PASSES, true, false>), g, b, 0, stream, p);
    } else {
        if (topk) hipLaunchKernelGGL((item_item_kernel<II_DENSE_PASSES, false, true>), g, b, 0, stream, p);
        else hipLaunchKernelGGL((item_item_kernel<II_DENSE_PASSES, false, false>), g, b, 0, stream, p);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int item_csr_pad() { return II_CSR_PAD; }
int item_item_chunk(bool sparse) { return (sparse ? II_SPARSE_PASSES : II_DENSE_PASSES) * II_PASS; }
int item_item_pass() { return II_PASS; }
int item_item_chunk_f64() { return II_F64_PASSES * II_PASS; }

// ------------------------------------------------------------------------------------
// Set-time helpers: CSR validation, per-chunk entry positions, magnitude bounds.
// ------------------------------------------------------------------------------------
// bad |= 1 for a column outside [0, n_cols), |= 2 for a row whose columns are not strictly ascending
__global__ void validate_csr_kernel(const int64_t *ptr, const int32_t *idx, int64_t n_rows, int n_cols, int need_sorted,
                                    int *bad)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    int flags = 0, prev = -1;
    for (int64_t q = ptr[r]; q < ptr[r + 1]; ++q) {
        const int c = idx[q];
        if (c < 0 || c >= n_cols) flags |= 1;
        if (need_sorted && c <= prev) flags |= 2;
        prev = c;
    }
    if (flags) atomicOr(bad, flags);
}

// cptr[row][c] = first entry of the row with column >= c * chunk (c = n_chunks: the row's end)
__global__ void build_chunk_ptr_kernel(const int64_t *ptr, const int32_t *idx, int64_t n_rows, int n_chunks, int chunk,
                                       int64_t *cptr)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows * (n_chunks + 1)) return;
    const int64_t r = t / (n_chunks + 1);
    const int c = (int)(t - r * (n_chunks + 1));
    int64_t lo = ptr[r], hi = ptr[r + 1];
    const int64_t target = (int64_t)c * chunk;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (idx[mid] < target) lo = mid + 1; else hi = mid; }
    cptr[t] = lo;
}

// values[q] *= col_scale[column of q]: the column scale of a CSR item matrix is folded into its values
__global__ void scale_csr_values_kernel(float *val, const int32_t *idx, int64_t nnz, const float *col_scale)
{
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nnz; q += (int64_t)gridDim.x * blockDim.x)
        val[q] = val[q] * col_scale[idx[q]];
}

int launch_scale_csr_values(float *val, const int32_t *idx, int64_t nnz, const float *col_scale, hipStream_t s)
{
    const int64_t blocks = (nnz + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks));
    hipLaunchKernelGGL(scale_csr_values_kernel, dim3(grid), dim3(256), 0, s, val, idx, nnz, col_scale);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// max |x| as the bits of a non-negative float (NaN sorts above +inf and is caught by the host)
__global__ void absmax_kernel(const float *x, int64_t n, uint32_t *out)
{
    uint32_t m = 0;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x)
        m = max(m, __float_as_uint(x[q]) & 0x7fffffffu);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// max over rows of sum |x| (fp64 sum, rounded up to fp32)
__global__ void row_l1_max_kernel(const int64_t *ptr, const float *val, int64_t n_rows, uint32_t *out)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t m = 0;
    if (r < n_rows) {
        double s = 0.0;
        for (int64_t q = ptr[r]; q < ptr[r + 1]; ++q) s += fabs((double)val[q]);
        m = __float_as_uint(__double2float_ru(s)) & 0x7fffffffu;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// (column, value) pairs of a CSR matrix: what item_sparse_topk_kernel streams
__global__ void interleave_csr_kernel(const int32_t *idx, const float *val, int64_t n, uint2 *pair)
{
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x)
        pair[q] = make_uint2((uint32_t)idx[q], __float_as_uint(val[q]));
}
int launch_interleave_csr(const int32_t *idx, const float *val, int64_t n, uint2 *pair, hipStream_t s)
{
    const int64_t blocks = (n + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks));
    hipLaunchKernelGGL(interleave_csr_kernel, dim3(grid), dim3(256), 0, s, idx, val, n, pair);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_validate_csr(const int64_t *ptr, const int32_t *idx, int64_t n_rows, int n_cols, int need_sorted, int *bad,
                        hipStream_t s)
{
    hipLaunchKernelGGL(validate_csr_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, s, ptr, idx, n_rows,
                       n_cols, need_sorted, bad);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_build_chunk_ptr(const int64_t *ptr, const int32_t *idx, int64_t n_rows, int n_chunks, int chunk, int64_t *cptr,
                           hipStream_t s)
{
    const int64_t total = n_rows * (n_chunks + 1);
    hipLaunchKernelGGL(build_chunk_ptr_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ptr, idx, n_rows,
                       n_chunks, chunk, cptr);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_absmax(const float *x, int64_t n, uint32_t *out, hipStream_t s)
{
    const int64_t blocks = (n + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks));
    hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, s, x, n, out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_row_l1_max(const int64_t *ptr, const float *val, int64_t n_rows, uint32_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(row_l1_max_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, s, ptr, val, n_rows, out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace ifr
