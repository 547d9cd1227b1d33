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
                    if (!TOPK && p.out64 != nullptr && col < p.n_items) p.out64[i * p.ld + col] = s;
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

    if (!TOPK) {
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int col = col0 + q * II_PASS + v;
                if (col < p.n_items) p.out[i * p.ld + col] = acc[q][v];
            }
        }
        return;
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
// CSR M, top-K: one workgroup per user with ONE running top-K list.  The user's catalogue is
// swept in PASSES of consecutive item chunks; a pass takes as many chunks as hold at most
// `sp_cap` (entry, exclusion) pairs of this user, found from the per-chunk entry positions
// (cptr) summed over the history rows.  A typical user (a few thousand nonzeros in all its
// history rows) is done in one to four passes instead of one per 4096-column chunk.
//   plan       : Tc[c] = entries of all history rows + exclusions with column < c * 4096;
//   hash pass  : the pass's entries are summed in an LDS hash table (open addressing, key =
//                column + 1, 64-bit fixed-point value as above; the user's exclusions are
//                entered first with a flag bit, entries that find a flagged key are dropped);
//   dense pass : a single chunk with more than sp_cap entries falls back to 4096 direct-mapped
//                64-bit accumulators + a touched bitmap in the same LDS;
//   select     : every thread converts its table slots to (orderable score, ~item) keys in
//                registers.  While the list is not full a lower bound of the K-th best comes from
//                K rounds of a DPP wave maximum over the threads' best keys; keys above the bound
//                (or above the list's K-th key) are appended behind a copy of the list and ranked
//                by counting (keys are unique, larger = better);
//   zeros      : columns no history row touches score exactly 0 * col_scale.  They only matter
//                when the list ends up short or holds scores <= 0; only then a second sweep marks
//                the touched columns per chunk and takes the K lowest untouched item ids.
// ------------------------------------------------------------------------------------
// make ABLATE=1 compiles the cumulative early exits (option "debug_mode" 101..105, timing only, tools/time_c5.py) into
// item_sparse_topk_kernel; the product build leaves them out (they cost SGPRs and a scratch slot)
#ifdef IFR_ABLATE
constexpr bool SP_DEV = true;
#else
constexpr bool SP_DEV = false;
#endif
#ifdef IFR_ABLATE
// dev: shader-clock cycles of thread 0 per phase, summed over workgroups (tools/time_c5.py prints them)
static __device__ unsigned long long sp_phase[16];
#define SP_STAMP(k) do { if (tid == 0 && (blockIdx.x & 63) == 0) { const long long now_ = clock64(); atomicAdd(&sp_phase[k], (unsigned long long)(now_ - sp_last)); sp_last = now_; } } while (0)
#define SP_COUNT(k) do { if (tid == 0 && (blockIdx.x & 63) == 0) atomicAdd(&sp_phase[k], 1ull); } while (0)
#else
#define SP_STAMP(k) do { } while (0)
#define SP_COUNT(k) do { } while (0)
#endif
constexpr int SP_CHUNK = II_SPARSE_PASSES * II_PASS;   // 4096 columns: granularity of cptr
constexpr int SP_BLOCK = 512;                           // threads per user: three workgroups (6 waves/SIMD, <= 85 VGPRs) per CU
constexpr int SP_SPT = 7;                               // hash slots per thread
constexpr int SP_SLOTS = SP_BLOCK * SP_SPT;             // 3584 slots (43 KB): the kernel is latency bound, occupancy beats table size
constexpr int SP_CAP = SP_SLOTS * 11 / 16;              // entries per hash pass (load <= 0.69; measured optimum 0.62-0.75)
constexpr int SP_RAW = SP_SLOTS * 3 / 2;                // 64-bit words: values, then 32-bit keys
constexpr int SP_DPT = (SP_CHUNK + SP_BLOCK - 1) / SP_BLOCK;   // dense pass: accumulators per thread
constexpr int SP_REG = SP_SPT > SP_DPT ? SP_SPT : SP_DPT;
constexpr int SP_HIST = 128;                            // history rows staged per round
constexpr int SP_EXCAP = 256;                           // exclusion row entries kept in LDS
constexpr int SP_MAXCH = 512;                           // item chunks (catalogue <= 2 097 152 items)
constexpr int SP_NG = SP_BLOCK / II_GROUP;
constexpr int SP_ZC = (SP_CHUNK + SP_BLOCK - 1) / SP_BLOCK;               // columns per thread in the zero sweep              // 16-lane groups per workgroup
constexpr uint32_t SP_FLAG = 0x80000000u;               // key flag: excluded column
static_assert(SP_RAW >= SP_CHUNK + 2 * (SP_CHUNK / 64) + 64, "dense pass accumulators + bitmaps must fit the table");

__device__ __forceinline__ uint32_t sp_float_key(float x)
{
    const uint32_t b = __float_as_uint(x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float sp_key_float(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}
__device__ __forceinline__ unsigned long long sp_key(float s, int col)
{
    return ((unsigned long long)sp_float_key(s) << 32) | (uint32_t)(~col);
}
__device__ __forceinline__ int sp_slot(int col)
{
    return (int)(((((uint32_t)col * 0x9E3779B1u) >> 16) * (uint32_t)SP_SLOTS) >> 16);
}
// Probe stride of a column (double hashing): 14 k + 1, k < 128 -- odd and = 1 mod 7, so coprime with SP_SLOTS = 2^9 * 7 and
// every probe sequence visits every slot.  With a stride of 1 (linear probing) the longest probe sequence among a wave's
// lanes -- which sets the number of probing rounds -- grows with the clusters; a key-dependent stride breaks them up.
static_assert(SP_SLOTS == 512 * 7, "sp_step assumes SP_SLOTS = 2^9 * 7");
__device__ __forceinline__ int sp_step(int col)
{
    const uint32_t h = ((uint32_t)col * 0x85EBCA6Bu) >> 25;          // 7 bits
    return (int)(h * 14u + 1u);
}
__device__ __forceinline__ int sp_next(int slot, int step)
{
    slot += step;
    return slot >= SP_SLOTS ? slot - SP_SLOTS : slot;
}
// maximum over the 64 lanes (DPP row shifts + row broadcasts), returned to every lane
__device__ __forceinline__ uint32_t sp_wave_max(uint32_t v)
{
#define SP_DPP_MAX(ctrl) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, 0xf, 0xf, false))
    SP_DPP_MAX(0x111);   // row_shr:1
    SP_DPP_MAX(0x112);   // row_shr:2
    SP_DPP_MAX(0x114);   // row_shr:4
    SP_DPP_MAX(0x118);   // row_shr:8   -> lane 15 of each row holds the row maximum
    SP_DPP_MAX(0x142);   // row_bcast:15
    SP_DPP_MAX(0x143);   // row_bcast:31 -> lane 63 holds the wave maximum
#undef SP_DPP_MAX
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// sum over each row of 16 lanes, valid in the row's last lane
__device__ __forceinline__ int sp_row_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    return v;
}

constexpr size_t SP_LDS_BYTES = (size_t)SP_RAW * 8 + (SP_MAXCH + 1) * 8 + 2 * SP_HIST * 8 + 32 * 8 + 2 * SP_HIST * 4 +
                                SP_EXCAP * 4 + (SP_BLOCK / 64) * 4 + 8;

__global__ __launch_bounds__(SP_BLOCK, 6) void item_sparse_topk_kernel(ItemItemParams p, int32_t *out_ids,
                                                                       float *out_scores)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_lds[];   // > 64 KB: dynamic
    unsigned long long *const raw = reinterpret_cast<unsigned long long *>(sp_lds);          // [SP_RAW] table / accumulators / merge keys
    unsigned long long *const Tc = raw + SP_RAW;                                             // [SP_MAXCH + 1]
    int64_t *const s0 = reinterpret_cast<int64_t *>(Tc + SP_MAXCH + 1);                      // [SP_HIST] current pass's segment
    int64_t *const s1 = s0 + SP_HIST;                                                        // [SP_HIST]   of each staged row
    unsigned long long *const top = reinterpret_cast<unsigned long long *>(s1 + SP_HIST);    // [32]
    int *const hj = reinterpret_cast<int *>(top + 32);                                       // [SP_HIST]
    float *const hx = reinterpret_cast<float *>(hj + SP_HIST);                               // [SP_HIST]
    int *const exl = reinterpret_cast<int *>(hx + SP_HIST);                                  // [SP_EXCAP]
    uint32_t *const wred = reinterpret_cast<uint32_t *>(exl + SP_EXCAP);                     // [SP_BLOCK / 64]
    int &s_nq = *reinterpret_cast<int *>(wred + SP_BLOCK / 64);
    int &s_cnt = *(&s_nq + 1);
    unsigned long long *const vals = raw;                                    // [SP_SLOTS]
    uint32_t *const keys = reinterpret_cast<uint32_t *>(raw + SP_SLOTS);     // [SP_SLOTS]
    unsigned long long *const qkey = raw;                                    // merge keys (table is in registers by then)
    uint32_t *const tbm = reinterpret_cast<uint32_t *>(raw + SP_RAW - 2 * (SP_CHUNK / 64));   // touched columns of a chunk
    uint32_t *const xbm = tbm + SP_CHUNK / 32;                                                // excluded columns of a chunk

    const int tid = threadIdx.x;
#ifdef IFR_ABLATE
    long long sp_last = clock64();
#endif
    const int lane = tid & 63, wave = tid >> 6;
    const int g = tid / II_GROUP, l = tid % II_GROUP;
    const int64_t i = blockIdx.x;
    const int u = p.user_ids[i];
    const int K = p.K;
    const int nc1 = p.n_chunks + 1;

    int64_t hb = 0, he = 0, eb = 0, ee = 0;
    if (u >= 0 && u < p.h_rows) { hb = p.h_ptr[u]; he = p.h_ptr[u + 1]; }
    if (p.ex_ptr != nullptr && u >= 0 && u < p.ex_rows) { eb = p.ex_ptr[u]; ee = p.ex_ptr[u + 1]; }
    const int64_t n_ex = ee - eb;
    const bool ex_lds = n_ex <= SP_EXCAP;
    const bool single = (he - hb) <= SP_HIST;           // the whole history fits one staging round
    const int m1 = single ? (int)(he - hb) : 0;

    for (int w = tid; w < SP_RAW; w += SP_BLOCK) raw[w] = 0ull;
    for (int w = tid; w < nc1; w += SP_BLOCK) Tc[w] = 0ull;
    if (ex_lds) for (int w = tid; w < (int)n_ex; w += SP_BLOCK) exl[w] = p.ex_idx[eb + w];
    if (tid < 32) top[tid] = 0ull;
    if (tid == 0) { s_nq = 0; s_cnt = 0; }

    // first exclusion entry with column >= col (every thread searches; block-uniform for uniform col)
    auto ex_lower_bound = [&](int col) -> int64_t {
        int64_t lo = 0, hi = n_ex;
        if (ex_lds) { while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (exl[mid] < col) lo = mid + 1; else hi = mid; } }
        else { while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (p.ex_idx[eb + mid] < col) lo = mid + 1; else hi = mid; } }
        return lo;
    };
    auto excluded = [&](int col) -> bool {
        const int64_t lo = ex_lower_bound(col);
        return lo < n_ex && (ex_lds ? exl[lo] : p.ex_idx[eb + lo]) == col;
    };
    // fn(col) for the user's exclusions with colA <= col < colB
    auto for_exclusions = [&](int colA, int colB, auto &&fn) {
        if (n_ex == 0) return;
        if (ex_lds) {
            for (int w = tid; w < (int)n_ex; w += SP_BLOCK) { const int c = exl[w]; if (c >= colA && c < colB) fn(c); }
        } else {
            for (int64_t w = ex_lower_bound(colA) + tid; w < n_ex; w += SP_BLOCK) {
                const int c = p.ex_idx[eb + w];
                if (c >= colB) break;
                fn(c);
            }
        }
    };
    // rows [base, base + m) of the history: item, weight and the entry range of chunks [ca, cb)
    auto stage = [&](int64_t base, int m, int ca, int cb, bool load_rows) {
        for (int e = tid; e < m; e += SP_BLOCK) {
            int j;
            if (load_rows) { j = p.h_idx[base + e]; hj[e] = j; hx[e] = p.h_val[base + e]; }
            else j = hj[e];
            const int64_t *cp = p.cptr + (int64_t)j * nc1;
            s0[e] = cp[ca];
            s1[e] = cp[cb];
        }
    };
    // fn(cols[4], vals[4], ok[4], x) over the staged rows' segments: a 16-lane group per row, four
    // strides of 16 entries in flight per lane
    auto sweep = [&](int m, auto &&fn) {
        const int gpr = m > 0 && m < SP_NG ? SP_NG / m : 1;     // lane groups per row (short histories)
        const int step = 4 * II_GROUP * gpr;
        for (int e = g / gpr; e < m; e += SP_NG / gpr) {
            if (gpr > 1 && g >= gpr * m) break;                   // groups left over
            const double x = ldexp((double)hx[e], p.fx_shift);
            const int64_t q0 = s0[e] + (g % gpr) * (4 * II_GROUP) + l;
            // entries left from this lane's first one; strides past the segment re-read its last entry of this
            // lane and arrive with ok = false
            int rem = (int)min(s1[e] - q0, (int64_t)0x7fffffff);
            const int32_t *ip = p.m_idx + q0;
            const float *vp = p.m_val + q0;
            for (; rem > 0; rem -= step, ip += step, vp += step) {
                int c[4];
                float w[4];
                bool ok[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    ok[t] = t * II_GROUP < rem;
                    const int o = min(t * II_GROUP, rem - 1);     // stay inside the segment's cache lines
                    c[t] = ip[o];
                    w[t] = vp[o];
                }
                fn(c, w, ok, x);
            }
        }
    };
    // every entry of the user's history rows with column in chunks [ca, cb); `pre` runs before the
    // first barrier (table set-up that the accumulation must see)
    auto walk = [&](int ca, int cb, auto &&fn) {
        if (single) {
            stage(hb, m1, ca, cb, false);
            __syncthreads();
            SP_STAMP(2);
            sweep(m1, fn);
        } else {
            for (int64_t base = hb; base < he; base += SP_HIST) {
                const int m = (int)min((int64_t)SP_HIST, he - base);
                __syncthreads();
                stage(base, m, ca, cb, true);
                __syncthreads();
                sweep(m, fn);
            }
        }
    };
    // x = history weight * 2^shift (exact); the product is exact (24 x 24 bits), rounded to nearest even
    auto fixed = [&](float wv, double x) -> unsigned long long {
        return (unsigned long long)__double2ll_rn(x * (double)wv);
    };
    // qkey[0, cnt) = the list, qkey[cnt, s_nq) = new keys -> the new list; call after a barrier
    auto rank_merge = [&]() {
        const int total = s_nq, cnt = s_cnt;
        if (total > cnt && !(SP_DEV && p.dbg == 105)) {
            for (int e0 = 0; e0 < total; e0 += SP_NG) {
                const int e = e0 + g;
                const unsigned long long k = e < total ? qkey[e] : ~0ull;
                int r = 0;
                for (int x = l; x < total; x += II_GROUP) r += qkey[x] > k ? 1 : 0;
                r = sp_row_sum(r);
                if (l == II_GROUP - 1 && e < total && r < K) top[r] = k;
            }
        }
        __syncthreads();
        for (int e = tid; e < total; e += SP_BLOCK) qkey[e] = 0ull;
        if (tid == 0) { const int c = min(K, total); s_cnt = c; s_nq = c; }
        if (total > SP_SLOTS) __syncthreads();                                  // zeroed words reach into the key area
    };

    // ---- plan: Tc[c] = entries + exclusions below the start of chunk c
    __syncthreads();
    SP_STAMP(0);
    for (int64_t base = hb; base < he; base += SP_HIST) {
        const int m = (int)min((int64_t)SP_HIST, he - base);
        if (base != hb) __syncthreads();
        for (int e = tid; e < m; e += SP_BLOCK) { hj[e] = p.h_idx[base + e]; hx[e] = p.h_val[base + e]; }
        __syncthreads();
        for (int e = g; e < m; e += SP_NG) {
            const int64_t *cp = p.cptr + (int64_t)hj[e] * nc1;
            const int64_t r0 = cp[0];
            for (int c = l + 1; c < nc1; c += II_GROUP) atomicAdd(&Tc[c], (unsigned long long)(cp[c] - r0));
        }
    }
    if (n_ex > 0) {
        __syncthreads();
        for (int c = tid + 1; c < nc1; c += SP_BLOCK)
            Tc[c] += (unsigned long long)ex_lower_bound(c >= (1 << 30) / SP_CHUNK ? 0x7fffffff : c * SP_CHUNK);
    }
    __syncthreads();
    SP_STAMP(1);
    SP_COUNT(9);

    if (SP_DEV && p.dbg == 101) return;                   // dev ablation 101..105: cumulative early exits, timing only
    // equal shares: ceil(total / cap) passes of about total / passes entries keep the table load even
    unsigned long long cap = (unsigned long long)p.sp_cap;
    {
        const unsigned long long total = Tc[p.n_chunks], np = (total + cap - 1) / cap;
        if (np > 1) cap = min(cap, (total + np - 1) / np + cap / 8);
    }
    int ca = 0;
    while (ca < p.n_chunks) {
        // ---- the pass: chunks [ca, cb)
        const unsigned long long t0 = Tc[ca];
        int cb = ca;
        for (;;) {
            const int ci = cb + 1 + lane;
            const bool fits = ci <= p.n_chunks && Tc[min(ci, p.n_chunks)] - t0 <= cap;
            const int n = __builtin_amdgcn_readfirstlane(__popcll(__ballot(fits)));   // Tc ascends: a prefix of lanes
            cb += n;
            if (n < 64) break;
        }
        const bool dense = cb == ca;
        if (dense) cb = ca + 1;
        if (Tc[cb] == t0) { ca = cb; continue; }          // nothing of this user in these chunks
        const int colA = ca * SP_CHUNK;
        const int colB = cb >= (1 << 30) / SP_CHUNK ? 0x7fffffff : cb * SP_CHUNK;

        unsigned long long kreg[SP_REG];
        if (!dense) {
            for_exclusions(colA, colB, [&](int col) {
                const uint32_t key = (uint32_t)(col + 1);
                int slot = sp_slot(col);
                const int step = sp_step(col);
                for (;;) {
                    const uint32_t old = atomicCAS(&keys[slot], 0u, key | SP_FLAG);
                    if (old == 0u || (old & ~SP_FLAG) == key) break;
                    slot = sp_next(slot, step);
                }
            });
            walk(ca, cb, [&](const int (&c)[4], const float (&w)[4], const bool (&ok)[4], double x) {
                if (SP_DEV && p.dbg == 102) { if (c[0] + c[1] + c[2] + c[3] + __float_as_int(w[0] + w[1] + w[2] + w[3]) == 0x7ffffff1) vals[0] = 1ull; return; }
                // An LDS atomic costs about the same with one active lane as with 64 (tools/micro/lds_sparse_lanes), so
                // the probing is arranged for few wave instructions: two queues of two entries per lane, every round
                // each lane probes once for the entry at the head of either queue (rounds = the longest queue's
                // probes, not the sum over the entries of the longest probe sequence among the lanes)
                // every round probes all four entries (finished ones with key 0: CAS(0 -> 0) changes nothing)
                uint32_t key[4];
                int slot[4];
                unsigned long long val[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    key[t] = ok[t] ? (uint32_t)(c[t] + 1) : 0u;
                    slot[t] = sp_slot(c[t]);
                    val[t] = fixed(w[t], x);
                }
                uint32_t old[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) { old[t] = 0u; if (key[t] != 0u) old[t] = atomicCAS(&keys[slot[t]], 0u, key[t]); }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const bool hit = old[t] == 0u || (old[t] & ~SP_FLAG) == key[t];
                    if (hit && key[t] != 0u && !(old[t] & SP_FLAG)) atomicAdd(&vals[slot[t]], val[t]);
                    key[t] = hit ? 0u : key[t];
                }
                // the entries whose slot belongs to another column walk on, one entry at a time: only their lanes run
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int step = sp_step(c[t]);
                    while (key[t] != 0u) {
                        slot[t] = sp_next(slot[t], step);
                        const uint32_t o = atomicCAS(&keys[slot[t]], 0u, key[t]);
                        if (o == 0u || (o & ~SP_FLAG) == key[t]) {
                            if (!(o & SP_FLAG)) atomicAdd(&vals[slot[t]], val[t]);
                            key[t] = 0u;
                        }
                    }
                }
            });
            __syncthreads();
            SP_STAMP(3);
            SP_COUNT(8);
            if (SP_DEV && p.dbg == 102) { ca = cb; continue; }
            if (SP_DEV && p.dbg == 103) {
                for (int v = 0; v < SP_SPT; ++v) { keys[tid + v * SP_BLOCK] = 0u; vals[tid + v * SP_BLOCK] = 0ull; }
                ca = cb;
                continue;
            }
            // ---- table slots -> sort keys in registers; the slots are left empty
            uint32_t kk[SP_SPT];
            unsigned long long av[SP_SPT];
#pragma unroll
            for (int v = 0; v < SP_SPT; ++v) {                                   // exchange: read and clear in one LDS instruction
                kk[v] = atomicExch(&keys[tid + v * SP_BLOCK], 0u);
                av[v] = atomicExch(&vals[tid + v * SP_BLOCK], 0ull);
            }
#pragma unroll
            for (int v = 0; v < SP_REG; ++v) {
                kreg[v] = 0ull;
                if (v >= SP_SPT) continue;
                const int col = (int)((kk[v] & ~SP_FLAG) - 1u);
                float s = (float)ldexp((double)(long long)av[v], -p.fx_shift);  // col_scale is folded into the values
                const bool live = kk[v] != 0u && !(kk[v] & SP_FLAG);
                if (live && s == 0.0f && p.pen != nullptr) s *= p.pen[col];     // 0 * inf = NaN like the reference
                if (s == 0.0f) s = 0.0f;                                        // -0 ranks as +0
                kreg[v] = live && s == s ? sp_key(s, col) : 0ull;
            }
        } else {
            for_exclusions(colA, colB, [&](int col) { atomicOr(&xbm[(col - colA) >> 5], 1u << (col & 31)); });
            walk(ca, cb, [&](const int (&c)[4], const float (&w)[4], const bool (&ok)[4], double x) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {                                   // an entry past the segment adds 0 / ors 0 at column 0
                    const int lc = ok[t] ? c[t] - colA : 0;
                    atomicAdd(&raw[lc], ok[t] ? fixed(w[t], x) : 0ull);
                    atomicOr(&tbm[lc >> 5], ok[t] ? 1u << (lc & 31) : 0u);
                }
            });
            __syncthreads();
#pragma unroll
            for (int v = 0; v < SP_REG; ++v) {
                kreg[v] = 0ull;
                if (v >= SP_DPT) continue;
                const int lc = tid + v * SP_BLOCK;
                if (lc >= SP_CHUNK) continue;
                const uint32_t bit = 1u << (lc & 31);
                if (!(tbm[lc >> 5] & bit)) continue;
                const long long a = (long long)raw[lc];
                raw[lc] = 0ull;
                if (xbm[lc >> 5] & bit) continue;
                const int col = colA + lc;
                float s = (float)ldexp((double)a, -p.fx_shift);
                if (s == 0.0f && p.pen != nullptr) s *= p.pen[col];
                if (s != s) continue;
                if (s == 0.0f) s = 0.0f;
                kreg[v] = sp_key(s, col);
            }
        }
        if (SP_DEV && p.dbg == 104) {
            if (dense) { __syncthreads(); if (tid < 2 * (SP_CHUNK / 32)) tbm[tid] = 0u; }
            unsigned long long z = 0;
            for (int v = 0; v < SP_REG; ++v) z ^= kreg[v];
            if (z == 0x1234567ull) top[0] = z;
            ca = cb;
            continue;
        }
        // ---- bound: the list's K-th key, or (list not full) the best wave's K-th best thread maximum
        const int cnt = s_cnt;
        if (cnt < K) {
            uint32_t lm = 0u;
#pragma unroll
            for (int v = 0; v < SP_REG; ++v) lm = max(lm, (uint32_t)(kreg[v] >> 32));
            uint32_t vk = 0u;
            for (int r = 0; r < K; ++r) {
                vk = sp_wave_max(lm);
                if (vk == 0u) break;
                const unsigned long long b = __ballot(lm == vk);
                if (lane == __ffsll((long long)b) - 1) lm = 0u;
            }
            if (lane == 0) wred[wave] = vk;
        }
        __syncthreads();                                   // every slot is in registers: raw becomes qkey
        SP_STAMP(4);
        unsigned long long tau;
        if (cnt < K) {
            uint32_t th = 0u;
#pragma unroll
            for (int w = 0; w < SP_BLOCK / 64; ++w) th = max(th, wred[w]);
            tau = th ? ((unsigned long long)th << 32) - 1ull : 0ull;
        } else tau = top[K - 1];
        if (dense && tid < 2 * (SP_CHUNK / 32)) tbm[tid] = 0u;                  // tbm and xbm are contiguous
        if (tid < cnt) qkey[tid] = top[tid];
#pragma unroll
        for (int v = 0; v < SP_REG; ++v) {
            const bool q = kreg[v] > tau;
            if (__any(q)) { if (q) qkey[atomicAdd(&s_nq, 1)] = kreg[v]; }
        }
        __syncthreads();
        SP_STAMP(5);
        rank_merge();
        ca = cb;
        SP_STAMP(6);
    }
    __syncthreads();
    if (SP_DEV && p.dbg >= 101) return;

    // ---- untouched columns (score 0 * col_scale): only when they can still enter the list
    {
        uint32_t *const zbm = reinterpret_cast<uint32_t *>(raw + SP_RAW - SP_CHUNK / 64);
        int found = 0;
        for (int chunk = 0; chunk < p.n_chunks && found < K; ++chunk) {
            const int c0 = chunk * SP_CHUNK;
            const int cnt = s_cnt;
            const unsigned long long tau = cnt == K ? top[K - 1] : 0ull;
            if (cnt == K && !(sp_key(0.0f, c0) > tau)) break;
            walk(chunk, chunk + 1, [&](const int (&c)[4], const float (&)[4], const bool (&ok)[4], double) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (ok[t]) { const int lc = c[t] - c0; atomicOr(&zbm[lc >> 5], 1u << (lc & 31)); }
            });
            __syncthreads();
            int mine = 0;
            uint32_t okbits = 0;
            for (int v = 0; v < SP_ZC; ++v) {
                const int lc = SP_ZC * tid + v;
                const int col = c0 + lc;
                bool ok = lc < SP_CHUNK && col < p.n_items && !((zbm[min(lc, SP_CHUNK - 1) >> 5] >> (lc & 31)) & 1u);
                if (ok && p.pen != nullptr) { const float z = 0.0f * p.pen[col]; ok = z == z; }
                if (ok) ok = sp_key(0.0f, col) > tau && !excluded(col);
                if (ok) { okbits |= 1u << v; ++mine; }
            }
            int incl = mine;                                                  // block exclusive prefix of `mine`
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if (lane >= off) incl += t; }
            if (lane == 63) wred[wave] = (uint32_t)incl;
            __syncthreads();
            int before = incl - mine, all = 0;
            for (int w = 0; w < SP_BLOCK / 64; ++w) { if (w < wave) before += (int)wred[w]; all += (int)wred[w]; }
            if (tid < SP_CHUNK / 32) zbm[tid] = 0u;
            if (tid < cnt) qkey[tid] = top[tid];
            before += found;
            for (int v = 0; v < SP_ZC && before < K; ++v)
                if (okbits & (1u << v)) { qkey[atomicAdd(&s_nq, 1)] = sp_key(0.0f, c0 + SP_ZC * tid + v); ++before; }
            found += all;
            __syncthreads();
            rank_merge();
            __syncthreads();
        }
    }

    const int cnt = s_cnt;
    if (tid < K) {
        const unsigned long long k = top[tid];
        out_ids[i * K + tid] = tid < cnt ? (int32_t)(~(uint32_t)k) : -1;
        out_scores[i * K + tid] = tid < cnt ? sp_key_float((uint32_t)(k >> 32)) : -INFINITY;
    }
    SP_STAMP(7);
}

// dev (make ABLATE=1): the phase counters, cleared by the read; -1 in a product build
int item_sparse_read_phases(unsigned long long *out16)
{
#ifdef IFR_ABLATE
    unsigned long long z[16] = {0};
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(out16, HIP_SYMBOL(sp_phase), sizeof(z)) != hipSuccess ||
        hipMemcpyToSymbol(HIP_SYMBOL(sp_phase), z, sizeof(z)) != hipSuccess) return -3;
    return 0;
#else
    (void)out16;
    return -1;
#endif
}

int launch_item_sparse_topk(const ItemItemParams &p, int32_t *out_ids, float *out_scores, hipStream_t stream)
{
    if (p.n <= 0 || p.n > (int64_t)0xffffffffu / SP_BLOCK || p.K < 1 || p.K > 32) return -1;
    if (p.n_chunks < 1 || p.n_chunks > SP_MAXCH || p.sp_cap < 1 || p.sp_cap > SP_CAP) return -1;
    static bool attr_done[64] = {false};                 // more than 64 KB of LDS needs the opt-in, once per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -3;
    bool &attr_set = attr_done[dev];
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(item_sparse_topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)SP_LDS_BYTES) != hipSuccess) return -3;
        attr_set = true;
    }
    hipLaunchKernelGGL(item_sparse_topk_kernel, dim3((unsigned)p.n), dim3(SP_BLOCK), SP_LDS_BYTES, stream, p, out_ids, out_scores);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int item_sparse_pass_capacity() { return SP_CAP; }
int item_sparse_max_chunks() { return SP_MAXCH; }

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
