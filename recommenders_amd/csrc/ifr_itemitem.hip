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
constexpr int II_GROUP = 16;                  // lanes that share one sparse row segment

typedef float f32x4 __attribute__((ext_vector_type(4)));


__device__ __forceinline__ bool ii_beats(float s, int id, float s2, int id2)
{
    return (s > s2) || (s == s2 && id < id2);
}

template <int PASSES, bool SPARSE, bool TOPK>
__global__ __launch_bounds__(II_BLOCK, 3) void item_item_kernel(ItemItemParams p)
{
    constexpr int CHUNK = PASSES * II_PASS;
    // sparse: int64 accumulators, later reused for the fp32 scores of the epilogue
    __shared__ __attribute__((aligned(16))) unsigned char buf[SPARSE ? CHUNK * 8 : (TOPK ? CHUNK * 4 : 16)];
    __shared__ uint32_t bm[TOPK ? CHUNK / 32 : 1];
    __shared__ int hj[II_HIST];
    __shared__ float hx[II_HIST];
    __shared__ int64_t sb[SPARSE ? II_HIST : 1], se[SPARSE ? II_HIST : 1];
    __shared__ float red_s[II_BLOCK / 64];
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
    if (p.pen != nullptr) {
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
            if (q < npass) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(p.pen + col0 + q * II_PASS);
                acc[q][0] *= w[0]; acc[q][1] *= w[1]; acc[q][2] *= w[2]; acc[q][3] *= w[3];
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
// CSR M, top-K: one workgroup per user sweeps the item chunks with ONE running top-K list.
// Per chunk the work is proportional to the nonzeros that fall into it, not to the chunk:
//   accumulate : 16-lane groups walk the chunk's segment of each history row; 64-bit fixed-point
//                LDS atomics as above; the first touch of a column (atomicOr on a bitmap) appends
//                it to a touched list;
//   drain      : only touched columns are converted, scaled, tested against the current K-th
//                best key and, if they pass, looked up in the exclusion row (LDS copy, binary
//                search); their accumulators are reset on the way, so the 32 KB are zeroed once;
//   merge      : qualifiers and the current list are ranked by counting (keys are unique:
//                (orderable score, ~item) packed in 64 bits, larger = better);
//   zeros      : untouched columns score exactly 0 * col_scale.  They can only matter while the
//                list is not full or its K-th key is not above the chunk's best possible zero,
//                and then only the K lowest item ids among them: a block prefix count picks those.
// The end of a row's chunk segment is the start of its next one, so each (history row, chunk)
// costs one pointer load, issued one chunk ahead.
// ------------------------------------------------------------------------------------
constexpr int SP_CHUNK = II_SPARSE_PASSES * II_PASS;   // 4096 columns, 32 KB of accumulators
constexpr int SP_HIST = 256;                            // history rows staged per round
constexpr int SP_SLAB = II_BLOCK;                       // touched columns drained per merge
constexpr int SP_EXCAP = 512;                           // exclusion row entries kept in LDS

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

__global__ __launch_bounds__(II_BLOCK, 3) void item_sparse_topk_kernel(ItemItemParams p, int32_t *out_ids,
                                                                       float *out_scores)
{
    __shared__ unsigned long long acc64[SP_CHUNK];
    __shared__ uint32_t tbm[SP_CHUNK / 32];
    __shared__ uint16_t tlist[SP_CHUNK];
    __shared__ int hj[SP_HIST];
    __shared__ float hx[SP_HIST];
    __shared__ int64_t s0[SP_HIST], s1[SP_HIST];        // current chunk's segment of each staged row
    __shared__ int exl[SP_EXCAP];
    __shared__ unsigned long long qkey[SP_SLAB + 32];
    __shared__ unsigned long long top[32];
    __shared__ int s_ntouch, s_nq, s_cnt;
    __shared__ int wsum[II_BLOCK / 64];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int g = tid / II_GROUP, l = tid % II_GROUP;
    const int64_t i = blockIdx.x;
    const int u = p.user_ids[i];
    const int K = p.K;
    const int nc1 = p.n_chunks + 1;

    int64_t hb = 0, he = 0, eb = 0, ee = 0;
    if (u >= 0 && u < p.h_rows) { hb = p.h_ptr[u]; he = p.h_ptr[u + 1]; }
    if (p.ex_ptr != nullptr && u >= 0 && u < p.ex_rows) { eb = p.ex_ptr[u]; ee = p.ex_ptr[u + 1]; }
    const int n_ex = (int)(ee - eb);
    const bool ex_lds = n_ex <= SP_EXCAP;
    const bool single = (he - hb) <= SP_HIST;           // the whole history fits one staging round

    for (int w = tid; w < SP_CHUNK; w += II_BLOCK) acc64[w] = 0ull;
    for (int w = tid; w < SP_CHUNK / 32; w += II_BLOCK) tbm[w] = 0u;
    if (ex_lds) for (int w = tid; w < n_ex; w += II_BLOCK) exl[w] = p.ex_idx[eb + w];
    if (tid < 32) top[tid] = 0ull;
    if (tid == 0) { s_ntouch = 0; s_nq = 0; s_cnt = 0; }
    if (single) {
        const int m = (int)(he - hb);
        for (int e = tid; e < m; e += II_BLOCK) {
            const int j = p.h_idx[hb + e];
            hj[e] = j;
            hx[e] = p.h_val[hb + e];
            const int64_t *cp = p.cptr + (int64_t)j * nc1;
            s0[e] = cp[0];
            s1[e] = cp[1];
        }
    }
    __syncthreads();

    auto excluded = [&](int col) -> bool {
        int lo = 0, hi = n_ex;
        if (ex_lds) {
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (exl[mid] < col) lo = mid + 1; else hi = mid; }
            return lo < n_ex && exl[lo] == col;
        }
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (p.ex_idx[eb + mid] < col) lo = mid + 1; else hi = mid; }
        return lo < n_ex && p.ex_idx[eb + lo] == col;
    };
    // qualifiers in qkey[0, s_nq) + the current list -> the new list (block-uniform control flow)
    auto merge = [&]() {
        const int nq = s_nq;
        if (nq == 0) return;
        const int cnt = s_cnt;
        if (tid < cnt) qkey[nq + tid] = top[tid];
        __syncthreads();
        const int total = nq + cnt;
        for (int e = tid; e < total; e += II_BLOCK) {
            const unsigned long long k = qkey[e];
            int rank = 0;
            for (int x = 0; x < total; ++x) rank += qkey[x] > k ? 1 : 0;
            if (rank < K) top[rank] = k;
        }
        __syncthreads();
        if (tid == 0) { s_cnt = min(K, total); s_nq = 0; }
        __syncthreads();
    };

    // In single mode the first 16 entries of the NEXT chunk's segment of up to two rows per lane
    // group are loaded one chunk ahead (pf_*), so a typical user (<= 32 history rows, segments of
    // <= 16 entries) never waits on HBM inside the chunk loop.  A prefetch may read past its
    // segment (never past the array); such entries are dropped by the q < q1 test at use.
    int pf_i[2] = {0, 0};
    float pf_v[2] = {0.f, 0.f};
    const int m1 = single ? (int)(he - hb) : 0;
    auto prefetch = [&](int (&ri)[2], float (&rv)[2], const int64_t *start) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int e = g + t * (II_BLOCK / II_GROUP);
            if (e < m1) {
                const int64_t q = start[e] + l;
                if (q < p.m_nnz) { ri[t] = p.m_idx[q]; rv[t] = p.m_val[q]; }
            }
        }
    };
    if (single) prefetch(pf_i, pf_v, s0);

    for (int chunk = 0; chunk < p.n_chunks; ++chunk) {
        const int c0 = chunk * SP_CHUNK;
        auto apply = [&](int ci, float wv, double x) {
            const int c = ci - c0;
            const double pr = x * (double)wv;                                  // exact: 24 x 24 bits
            const long long v = __double2ll_rn(ldexp(pr, p.fx_shift));          // nearest even
            atomicAdd(&acc64[c], (unsigned long long)v);
            const uint32_t bit = 1u << (c & 31);
            if (!(atomicOr(&tbm[c >> 5], bit) & bit)) tlist[atomicAdd(&s_ntouch, 1)] = (uint16_t)c;
        };
        // ---- accumulate
        if (single) {
            const int m = m1;
            int nf_i[2] = {0, 0};
            float nf_v[2] = {0.f, 0.f};
            int64_t nxt = 0;
            if (chunk + 1 < p.n_chunks) {
                prefetch(nf_i, nf_v, s1);
                // the boundary after the NEXT chunk, in flight while this chunk is accumulated
                if (tid < m) nxt = p.cptr[(int64_t)hj[tid] * nc1 + chunk + 2];
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int e = g + t * (II_BLOCK / II_GROUP);
                if (e < m) {
                    const double x = (double)hx[e];
                    const int64_t q1 = s1[e];
                    int64_t q = s0[e] + l;
                    if (q < q1) apply(pf_i[t], pf_v[t], x);
                    for (q += II_GROUP; q < q1; q += II_GROUP) apply(p.m_idx[q], p.m_val[q], x);
                }
            }
            for (int e = g + 2 * (II_BLOCK / II_GROUP); e < m; e += II_BLOCK / II_GROUP) {
                const double x = (double)hx[e];
                const int64_t q1 = s1[e];
                for (int64_t q = s0[e] + l; q < q1; q += II_GROUP) apply(p.m_idx[q], p.m_val[q], x);
            }
            __syncthreads();
            if (tid < m) { s0[tid] = s1[tid]; s1[tid] = nxt; }
            pf_i[0] = nf_i[0]; pf_i[1] = nf_i[1]; pf_v[0] = nf_v[0]; pf_v[1] = nf_v[1];
        } else {
            for (int64_t base = hb; base < he; base += SP_HIST) {
                const int m = (int)min((int64_t)SP_HIST, he - base);
                __syncthreads();
                for (int e = tid; e < m; e += II_BLOCK) {
                    const int j = p.h_idx[base + e];
                    hj[e] = j;
                    hx[e] = p.h_val[base + e];
                    const int64_t *cp = p.cptr + (int64_t)j * nc1 + chunk;
                    s0[e] = cp[0];
                    s1[e] = cp[1];
                }
                __syncthreads();
                for (int e = g; e < m; e += II_BLOCK / II_GROUP) {
                    const double x = (double)hx[e];
                    const int64_t q1 = s1[e];
                    for (int64_t q = s0[e] + l; q < q1; q += II_GROUP) apply(p.m_idx[q], p.m_val[q], x);
                }
            }
        }
        __syncthreads();
        // ---- drain the touched columns, one slab per merge
        const int nt = s_ntouch;
        for (int sbase = 0; sbase < nt; sbase += SP_SLAB) {
            const int cnt = s_cnt;
            const unsigned long long tau = cnt == K ? top[K - 1] : 0ull;
            const int e = sbase + tid;
            if (e < nt) {
                const int lc = tlist[e];
                const long long a = (long long)acc64[lc];
                acc64[lc] = 0ull;
                const int col = c0 + lc;
                float s = (float)ldexp((double)a, -p.fx_shift);
                if (p.pen != nullptr) s *= p.pen[col];
                if (s == s) {
                    if (s == 0.0f) s = 0.0f;                                   // -0 ranks as +0
                    const unsigned long long k = sp_key(s, col);
                    if (k > tau && !excluded(col)) qkey[atomicAdd(&s_nq, 1)] = k;
                }
            }
            __syncthreads();
            merge();
        }
        // ---- untouched columns (score 0 * col_scale), only while they can still enter the list
        {
            const int cnt = s_cnt;
            const unsigned long long tau = cnt == K ? top[K - 1] : 0ull;
            if (cnt < K || sp_key(0.0f, c0) > tau) {
                int mine = 0;
                uint32_t okbits = 0;
#pragma unroll 4
                for (int v = 0; v < 16; ++v) {
                    const int lc = 16 * tid + v;
                    const int col = c0 + lc;
                    bool ok = col < p.n_items && !((tbm[lc >> 5] >> (lc & 31)) & 1u);
                    if (ok && p.pen != nullptr) { const float z = 0.0f * p.pen[col]; ok = z == z; }
                    if (ok) ok = sp_key(0.0f, col) > tau && !excluded(col);
                    if (ok) { okbits |= 1u << v; ++mine; }
                }
                int incl = mine;                                              // block exclusive prefix of `mine`
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if (lane >= off) incl += t; }
                if (lane == 63) wsum[wave] = incl;
                __syncthreads();
                int before = incl - mine;
                for (int w = 0; w < wave; ++w) before += wsum[w];
                for (int v = 0; v < 16 && before < K; ++v)
                    if (okbits & (1u << v)) { qkey[atomicAdd(&s_nq, 1)] = sp_key(0.0f, c0 + 16 * tid + v); ++before; }
                __syncthreads();
                merge();
            }
        }
        // ---- reset the touched bitmap and list for the next chunk
        for (int w = tid; w < SP_CHUNK / 32; w += II_BLOCK) tbm[w] = 0u;
        if (tid == 0) s_ntouch = 0;
        __syncthreads();
    }

    const int cnt = s_cnt;
    if (tid < K) {
        const unsigned long long k = top[tid];
        out_ids[i * K + tid] = tid < cnt ? (int32_t)(~(uint32_t)k) : -1;
        out_scores[i * K + tid] = tid < cnt ? sp_key_float((uint32_t)(k >> 32)) : -INFINITY;
    }
}

int launch_item_sparse_topk(const ItemItemParams &p, int32_t *out_ids, float *out_scores, hipStream_t stream)
{
    if (p.n <= 0 || p.n > (int64_t)0xffffffffu / II_BLOCK || p.K < 1 || p.K > 32) return -1;
    hipLaunchKernelGGL(item_sparse_topk_kernel, dim3((unsigned)p.n), dim3(II_BLOCK), 0, stream, p, out_ids, out_scores);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_item_item(const ItemItemParams &p, bool sparse, bool topk, hipStream_t stream)
{
    // HIP limits grid.x * block.x to 32 bits and grid.y to 65535
    if (p.n <= 0 || p.n_chunks <= 0 || p.n > (int64_t)0xffffffffu / II_BLOCK || p.n_chunks > 65535) return -1;
    const dim3 g((unsigned)p.n, (unsigned)p.n_chunks), b(II_BLOCK);
    if (sparse) {
        if (topk) return -1;                             // CSR top-K is item_sparse_topk_kernel
        hipLaunchKernelGGL((item_item_kernel<II_SPARSE_PASSES, true, false>), g, b, 0, stream, p);
    } else {
        if (topk) hipLaunchKernelGGL((item_item_kernel<II_DENSE_PASSES, false, true>), g, b, 0, stream, p);
        else hipLaunchKernelGGL((item_item_kernel<II_DENSE_PASSES, false, false>), g, b, 0, stream, p);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int item_item_chunk(bool sparse) { return (sparse ? II_SPARSE_PASSES : II_DENSE_PASSES) * II_PASS; }
int item_item_pass() { return II_PASS; }

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
