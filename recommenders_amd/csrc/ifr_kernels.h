// ifr_kernels.h -- CDNA4 (gfx950) kernels of the scoring + ranking-evaluation path.
//
// Layout in HBM (owned by the handle, see ifr_api.hip):
//   U      [n_users][KD]        fp32, KD = k padded to 32, 64, 96, ... 256 with zeros
//   V      [n_tiles*32 + 32][KD] fp32, zero rows past n_items (one spare tile so the
//                                prefetch of "tile + 1" never leaves the allocation)
//   ibase  [n_tiles*32 + 32]    fp32 item-side base (bias), zeros if the model has none
//   ubase  [n_users]            fp32 user-side base
//   exclusions as per-user lists of (tile, 32-bit mask) stored in place of the CSR row:
//          entries of user u live at ex_ptr[u] .. ex_ptr[u] + ex_cnt[u]
//
// The fused kernel computes a 32(items) x 32(users) score block per wave with
// v_mfma_f32_32x32x2_f32: A = item factors read from an LDS tile, B = user factors held in
// registers for the whole item sweep.  In the accumulator a lane owns ONE user (column
// lane&31) and 16 item rows, so the per-user running top-K is lane-private: no cross-lane
// traffic in the select.  Scores never leave the CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ifr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE_I = 32;        // items per LDS tile (one MFMA block)
constexpr int WAVES = 4;          // waves per workgroup
constexpr int BLOCK = 256;
constexpr int USERS_PER_BLOCK = 128;
constexpr int PEND = 16;          // pending-candidate slots per lane (LDS)
constexpr int NO_TILE = 0x7fffffff;
constexpr int WIDE_AGE = 64;        // WIDE variant: tiles a pending entry may wait for its flush
constexpr int WIDE_MAX_LISTS = 12;  // WIDE variant: lists per user (2 per item chunk)

struct TopkParams {
    const float *U, *V, *ibase, *ubase;
    const int32_t *user_ids; // rows of U; ids outside [0, n_users) are clamped into it (no out-of-bounds read; the
                             // host mirror rejects them before the call)
    int64_t n_users;
    int64_t n;               // users in this batch
    int n_items;
    int n_tiles;             // ceil(n_items / 32)
    int chunk_tiles;         // tiles per item chunk
    int n_chunks;
    // Two-region grid (tail balancing): user tiles [0, n_utA) run n_chunks item chunks each, in whole rounds of the
    // resident workgroup slots; the remaining user tiles run n_chunksB >= n_chunks shorter chunks, sized so that
    // they fill ONE last round -- instead of a last round of full-length workgroups on a fraction of the CUs.
    // Workgroups [0, gridA) belong to region A.  One region: n_utA = all user tiles, gridA = the whole grid.
    int n_utA, gridA;
    int chunk_tilesB, n_chunksB;
    int64_t part_offB;       // first entry of region B's partial lists in part_scores / part_ids (region A: 2 * n_chunks
                             // lists of KL entries per user from entry 0; region B: 2 * n_chunksB lists per user from here)
    int K;                   // requested list length (<= KL)
    const int64_t *ex_ptr;   // may be null
    const int32_t *ex_cnt;
    const uint2 *ex_pair;    // (tile, 32-bit mask) entries of a user at ex_ptr[u] .. ex_ptr[u] + ex_cnt[u]
    int64_t ex_rows;
    float *part_scores;      // [n_slots][n_chunks*2][KL]
    int32_t *part_ids;
    uint32_t *lb;            // [n_slots padded to 128] per-user lower-bound keys, zeroed per call (may be null)
    // WIDE variant (16 < K <= 128): the lane-private lists only yield bounds; every candidate that can still reach the
    // user's top K is appended to the user's buffer in global memory and selected exactly afterwards
    uint32_t *lbm;           // [n][wide_lists] per (user, list) word: key of the list's m_rank-th best score (0 = not that many yet)
    int m_rank;              // ceil(K / wide_lists); 17 when the lists cannot hold K entries together (no bound from them)
    uint32_t *started;       // [user tiles] workgroups of the user tile that have begun (zeroed per call)
    uint32_t *bw;            // [n] per-user bound word: key of the best certified lower bound of the K-th best (0 = none)
    int wide_lists;          // lists per user = 2 * n_chunks
    uint2 *cand;             // [n][cand_cap] (score bits, item id), unordered
    uint32_t *cand_cnt;      // [n] entries appended (may exceed cand_cap: overflow, the host falls back for that user)
    int cand_cap;
    float *dense;            // STORE variant: also write every score, dense[slot][item] (may be null otherwise)
    int64_t dense_ld;        // floats per row, >= n_tiles * 32 and a multiple of 4
    unsigned long long *stamps;  // dev-only: [grid][4 waves][8] cycle sums per phase (debug bit 7), else null
    int dbg;                 // dev-only ablation switch, bits 0-2: 0 = normal, 2 = never hits, 3 = also no staging and no
                             // barrier, 4 = no staging; bit 6 (host side): extra LDS so that one workgroup fits per CU
};

// item row inside a 32x32 accumulator block held by register r of a lane in half h
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// float <-> order-preserving uint32 key (0 is below every float: "no bound yet")
__device__ __forceinline__ uint32_t float_to_key(float x)
{
    const uint32_t b = __float_as_uint(x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(uint32_t k)
{
    if (k == 0) return -INFINITY;
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}
// largest float strictly below x (so that  s > next_below(x)  <=>  s >= x); -inf and NaN map to -inf
__device__ __forceinline__ float next_below(float x)
{
    if (!(x > -INFINITY)) return -INFINITY;
    uint32_t b = __float_as_uint(x);
    if ((b << 1) == 0) return __uint_as_float(0x80000001u);      // +-0 -> -denorm_min
    b = (b & 0x80000000u) ? b + 1 : b - 1;
    return __uint_as_float(b);
}

// Sorted insertion into a lane-private descending list.  Strict '>' keeps earlier (smaller
// item index) entries ahead of later equal scores, which is the canonical tie order because a
// lane sees its items in ascending index order.
// N <= KL: only the first N entries take part (a list that never needs more than N keeps the rest at -inf).
template <int KL, int N = KL>
__device__ __forceinline__ void list_insert(float (&ls)[KL], int (&li)[KL], float x, int id)
{
    static_assert(N >= 1 && N <= KL, "list_insert: N entries of a KL list");
#pragma unroll
    for (int i = N - 1; i >= 1; --i) {
        const bool ci = x > ls[i];
        const bool cim = x > ls[i - 1];
        const int take = cim ? li[i - 1] : id;
        li[i] = ci ? take : li[i];
        ls[i] = __builtin_amdgcn_fmed3f(x, ls[i - 1], ls[i]);
    }
    li[0] = (x > ls[0]) ? id : li[0];
    ls[0] = fmaxf(x, ls[0]);
}

// ls[k] for a wave-uniform k in [LO, HI): a tree of scalar branches down to one register move.  (Written as a min
// chain with '(i < K)' selects, hipcc keeps one loop-invariant SGPR mask pair per entry, spills them to VGPR lanes
// and reads them back around every select: ~90 instructions per call; an 'i == K-1' select chain becomes a
// scratch-indexed load.)  The empty asm keeps each leaf a block of its own.
template <int LO, int HI, int KL>
__device__ __forceinline__ float list_pick(const float (&ls)[KL], int k)
{
    if constexpr (HI - LO == 1) {
        float t = ls[LO];
        asm volatile("" : "+v"(t));
        return t;
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (k < MID) return list_pick<LO, MID, KL>(ls, k);
        return list_pick<MID, HI, KL>(ls, k);
    }
}

template <int KL>
__device__ __forceinline__ float list_tau(const float (&ls)[KL], int K)
{
    // ls is descending, so the K-th best is entry K - 1 (K >= KL: the last entry)
    return list_pick<0, KL, KL>(ls, min(max(K, 1), KL) - 1);
}

// ------------------------------------------------------------------------------------
// Fused score + exclusion + per-user top-K over one (user tile, item chunk).
// grid.x = n_chunks * n_user_tiles, chunk-major so that concurrently resident workgroups
// stream the same item chunk through L2.
// ------------------------------------------------------------------------------------
// STORE: the scores are also written out as dense rows (the same values ifr_score_dense computes), so that the
// row-wise consumers (AUC, list metrics) cost one HBM pass instead of a second GEMM.  A user's 32 scores of a
// tile are one 128-byte line, written as 8 x 16 bytes by the user's two half-lanes.
// WIDE: cut-offs 16 < K <= 128.  A lane still keeps a sorted 16-list in registers, but only as a source of bounds: list
// i of a user (one per half-lane and item chunk, L lists in all) publishes its m-th best score, m = ceil(K / L); the
// L lists hold distinct items, so at least K items of the user score >= B = min_i (m-th best of list i), i.e. B is a
// lower bound of the user's K-th best.  B goes to the user's bound word by atomicMax, and every lane re-reads that
// word once per tile (the load has the whole tile to come back).  Every non-excluded score >= the bound is appended
// to the user's candidate buffer in global memory (one returning atomic per lane and flush reserves the slots);
// select_candidates_kernel then picks the exact top K.  Bounds only rise, a stale or missing word only costs
// candidates, never a result.  Workgroups are numbered user-tile-major here so that the chunks of a user tile start
// together and their words meet early; each takes every n_chunks-th tile, so that every list samples the whole
// catalogue.  (Tried: the exact K-th best of the union of the lists from four published ranks per list -- 20 % fewer
// candidates, but the 16-byte loads and 16 L compares per flush cost far more than they saved.)
template <int KD, int KL, bool HAS_BASE, bool STORE = false, bool WIDE = false>
__global__ __launch_bounds__(BLOCK, KD > 128 ? 1 : 2) void score_topk_kernel(TopkParams p)
{
    constexpr int ROW = KD + 4;                 // padded LDS row (floats); +4 => conflict-free b128 reads
    constexpr int C4 = KD / 4;                  // float4 chunks per factor row
    constexpr int STAGE = (TILE_I * C4) / BLOCK; // float4 loads per thread per tile
    static_assert((TILE_I * C4) % BLOCK == 0, "tile must split evenly");
    __shared__ __attribute__((aligned(16))) float vt[2][TILE_I][ROW];
    // item bases of tile t live in ibt[t & 1].  They are stored at the TOP of iteration t (from the register the
    // prefetch of iteration t - 1 filled), not with the tile's factors at the end of iteration t - 1: the select of an
    // iteration reads the OTHER slot all through its MFMA zone, and a store at the end of the previous iteration let a
    // fast wave overwrite what a slower wave of the workgroup still read (round 1's latent race, seen with k <= 32).
    __shared__ __attribute__((aligned(16))) float ibt[2][TILE_I];
    __shared__ uint2 pend[PEND + 1][BLOCK];      // row PEND: dump row of the straight-line append

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int n_ut = (int)((p.n + USERS_PER_BLOCK - 1) / USERS_PER_BLOCK);
    int chunk, ut, chunk_tiles, lists;
    int64_t part_off = 0, slot0 = 0;
    if (WIDE) {
        // user-tile-major: the chunks of a user tile are consecutive workgroups, dispatched together, so that the lists
        // of a user fill side by side (speed only: any placement gives the same result)
        ut = blockIdx.x / p.n_chunks;
        chunk = blockIdx.x - ut * p.n_chunks;
        chunk_tiles = p.chunk_tiles;
        lists = 2 * p.n_chunks;
    } else if ((int)blockIdx.x < p.gridA) {
        chunk = blockIdx.x / p.n_utA;
        ut = blockIdx.x - chunk * p.n_utA;
        chunk_tiles = p.chunk_tiles;
        lists = 2 * p.n_chunks;
    } else {
        const int b = blockIdx.x - p.gridA, n_utB = n_ut - p.n_utA;
        chunk = b / n_utB;
        ut = p.n_utA + (b - chunk * n_utB);
        chunk_tiles = p.chunk_tilesB;
        lists = 2 * p.n_chunksB;
        part_off = p.part_offB;
        slot0 = (int64_t)p.n_utA * USERS_PER_BLOCK;
    }
    // Tiles of this workgroup: t0, t0 + ts, ... < t1.  Non-WIDE: a contiguous chunk (ts = 1).  WIDE: every n_chunks-th
    // tile, so that each list sees a sample of the whole catalogue -- with contiguous chunks a catalogue sorted by
    // score (popularity order) would leave the min-over-lists bound at the worst chunk's level.
    const int ts = WIDE ? p.n_chunks : 1;
    const int t0 = WIDE ? chunk : chunk * chunk_tiles;
    const int t1 = WIDE ? t0 + ((p.n_tiles - chunk + ts - 1) / ts) * ts : min(t0 + chunk_tiles, p.n_tiles);

    const int64_t slot = (int64_t)ut * USERS_PER_BLOCK + wave * 32 + j;
    const bool live = slot < p.n;
    const int uid = (int)min((int64_t)max(p.user_ids[live ? slot : p.n - 1], 0), p.n_users - 1);

    if (WIDE && p.n_chunks > 1) {
        // Rendezvous of the user tile's workgroups.  Until every list of a user has published, the bound is -inf and a
        // lane appends every score it sees; workgroups are dispatched in block order, but once the chip is full the
        // next one only starts when a slot frees -- measured: siblings 30-60 tiles (and at a round boundary a whole
        // workgroup's time) apart, 2-4 x the candidates.  So a workgroup waits for its siblings before its first
        // tile.  They are the next blocks in dispatch order, so every other resident workgroup runs on and frees
        // slots; the spin is bounded all the same (no ordering is promised): on a time-out the workgroup just
        // proceeds with more candidates.
        if (tid == 0) {
            atomicAdd(&p.started[ut], 1u);
            for (int spins = 0; spins < 200000; ++spins) {
                if (__hip_atomic_load(&p.started[ut], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (uint32_t)p.n_chunks) break;
                __builtin_amdgcn_s_sleep(64);
            }
        }
        __syncthreads();
    }

    // user factors: lane (j,h) keeps U[uid][8g + 4h + e], g < KD/8, e < 4
    f32x4 ureg[KD / 8];
    {
        const float *up = p.U + (int64_t)uid * KD + 4 * h;
#pragma unroll
        for (int g = 0; g < KD / 8; ++g) ureg[g] = *reinterpret_cast<const f32x4 *>(up + 8 * g);
    }
    const float ub = HAS_BASE ? p.ubase[uid] : 0.0f;
    // STORE: row of this lane's user; the idle lanes of the last user tile recompute user n-1 and rewrite its values
    float *drow = STORE ? p.dense + (live ? slot : p.n - 1) * p.dense_ld + 4 * h : nullptr;

    // exclusion cursor: first (tile, mask) entry with tile >= t0
    // (nt, nm) is the next entry, (nt2, nm2) the one after it: an entry is loaded at least one
    // tile before it is needed, so advancing the cursor never waits on memory inside the sweep
    int64_t ex_cur = 0, ex_end = 0;
    int nt = NO_TILE, nt2 = NO_TILE;
    uint32_t nm = 0, nm2 = 0;
    if (p.ex_ptr != nullptr && uid < p.ex_rows) {
        const int64_t b = p.ex_ptr[uid];
        int lo = 0, hi = p.ex_cnt[uid];
        ex_end = b + hi;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((int)p.ex_pair[b + mid].x < t0) lo = mid + 1; else hi = mid;
        }
        ex_cur = b + lo;
        if (ex_cur < ex_end) { const uint2 e = p.ex_pair[ex_cur]; nt = (int)e.x; nm = e.y; }
        if (ex_cur + 1 < ex_end) { const uint2 e = p.ex_pair[ex_cur + 1]; nt2 = (int)e.x; nm2 = e.y; }
    }

    float ls[KL];
    int li[KL];
#pragma unroll
    for (int i = 0; i < KL; ++i) { ls[i] = -INFINITY; li[i] = NO_TILE; }
    float tau = -INFINITY;
    int cnt = 0;
    // A lower bound L on the user's K-th best score (from the partner half-lane or from another
    // item chunk's workgroup, via p.lb) prunes with '>=': s > next_below(L).  The lane's own K-th
    // best prunes with '>' (its list only holds earlier, smaller item ids).
    // (a region with ONE item chunk per user tile has nobody to share with: the word is neither read nor raised)
    const bool share_lb = !WIDE && p.lb != nullptr && lists > 2;
    if (share_lb) tau = next_below(key_to_float(p.lb[slot]));
    if (WIDE && !live) tau = INFINITY;          // idle lanes of the last user tile must not append for user n - 1
    uint32_t bw_seen = 0;
    int age = 0;                                // WIDE: tiles since the last flush                       // WIDE: the user's bound word as loaded one tile ago
    auto flush_wide = [&]() __attribute__((always_inline)) {
        // every pending entry enters the lane's list; those that still reach the bound are the user's candidates
        // (rounds while some lane still holds an entry: one vote per round instead of a six-step wave maximum)
        int keep = 0;
        for (int q = 0; __any(q < cnt); ++q) {
            const uint2 e = pend[q][tid];
            const float x = (q < cnt) ? __uint_as_float(e.x) : -INFINITY;     // -inf never inserts
            list_insert<KL>(ls, li, x, (int)e.y);
            keep += x > tau ? 1 : 0;
        }
        if (keep > 0) {
            const uint32_t pos0 = atomicAdd(&p.cand_cnt[slot], (uint32_t)keep);
            uint2 *dst = p.cand + slot * (int64_t)p.cand_cap;
            uint32_t w = pos0;
            for (int q = 0; q < cnt; ++q) {
                const uint2 e = pend[q][tid];
                if (__uint_as_float(e.x) > tau) { if (w < (uint32_t)p.cand_cap) dst[w] = e; ++w; }
            }
        }
        cnt = 0;
        age = 0;
        if (live) {
            uint32_t *words = p.lbm + slot * (int64_t)p.wide_lists;
            // fewer than m entries yet (-inf), or lists too few to hold K entries together (m_rank > KL): "no bound", key 0
            const float own = list_tau<KL>(ls, p.m_rank);
            const uint32_t pub = (p.m_rank <= KL && own > -INFINITY) ? float_to_key(own) : 0u;
            __hip_atomic_store(words + chunk * 2 + h, pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t mn = pub;
            for (int i = 0; i < p.wide_lists; i += 4) {
                const uint32_t a = __hip_atomic_load(words + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t b = __hip_atomic_load(words + i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t c = i + 2 < p.wide_lists ? __hip_atomic_load(words + i + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0u;
                const uint32_t d = i + 3 < p.wide_lists ? __hip_atomic_load(words + i + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0u;
                mn = min(min(mn, a), min(b, min(c, d)));
            }
            float B = key_to_float(mn);
            if (p.K <= 2 * KL) {
                // K <= 32: the two half-lanes of the user hold K entries between them -- a bound that needs no other workgroup
                const float half = list_tau<KL>(ls, (p.K + 1) >> 1);
                B = fmaxf(B, fminf(half, __shfl_xor(half, 32)));
            }
            if (B > -INFINITY) atomicMax(&p.bw[slot], float_to_key(B));            // for the user's other lanes (per-tile refresh)
            tau = fmaxf(tau, next_below(B));                                         // s > tau  <=>  s >= B
        }
    };
    // full: some lane holds PEND entries (the overflow path), so the rounds are PEND; otherwise they run while a vote
    // finds a lane with an entry left (a wave maximum by six bpermute steps cost ~700 cycles per flush)
    auto flush = [&](const bool full) __attribute__((always_inline)) {
        if (WIDE) { flush_wide(); return; }
        // the user's shared bound as the other item chunks left it: loaded now, used after the insertion rounds (any
        // value the word ever held is a valid bound, so the older one costs nothing and hides the round trip)
        uint32_t lb_seen = 0;
        if (share_lb) lb_seen = __hip_atomic_load(p.lb + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the rounds read one entry ahead (the LDS latency of round q + 1 behind the insertion of round q);
        // K <= 10 (the metric's cut-off): entries 10 ... KL - 1 can never reach the result, the insertion skips them
        uint2 e = pend[0][tid];
        if (KL > 10 && p.K <= 10) {
#pragma nounroll
            for (int q = 0; full ? q < PEND : __any(q < cnt); ++q) {
                const uint2 en = pend[min(q + 1, PEND)][tid];
                const float x = (q < cnt) ? __uint_as_float(e.x) : -INFINITY;     // -inf never inserts
                list_insert<KL, (KL > 10 ? 10 : KL)>(ls, li, x, (int)e.y);
                e = en;
            }
        } else {
#pragma nounroll
            for (int q = 0; full ? q < PEND : __any(q < cnt); ++q) {
                const uint2 en = pend[min(q + 1, PEND)][tid];
                const float x = (q < cnt) ? __uint_as_float(e.x) : -INFINITY;
                list_insert<KL>(ls, li, x, (int)e.y);
                e = en;
            }
        }
        cnt = 0;
        // Lower bounds on the user's K-th best score: this lane's K-th best, the partner half-lane's,
        // and min(ceil(K/2)-th best of both halves) -- the two lists hold distinct items, so at least K
        // of the user's items reach it; for two random halves it is close to the K-th best of their
        // union, about twice as selective as either K-th best alone.
        // K > KL (the pair path, 16 < K <= 32): the list cannot certify the user's K-th best; its last entry is only
        // this list's own admission threshold.  What it drops is caught at the merge (a full list whose last entry
        // ranks before the user's K-th: that user is ranked again from a dense row).
        const float own = list_tau<KL>(ls, p.K);
        const float own_c = p.K <= KL ? own : -INFINITY;
        const float ownh = list_tau<KL>(ls, (p.K + 1) >> 1);
        // the pair bound needs the two lists to hold K entries between them
        const float pairb = ((p.K + 1) >> 1) <= KL ? fminf(ownh, __shfl_xor(ownh, 32)) : -INFINITY;
        float ext = fmaxf(__shfl_xor(own_c, 32), pairb);
        if (share_lb) {
            const uint32_t mine = float_to_key(fmaxf(own_c, ext));
            if (mine > lb_seen) atomicMax(&p.lb[slot], mine);          // result unused: no round trip to wait for
            ext = fmaxf(ext, key_to_float(lb_seen));
        }
        tau = fmaxf(fmaxf(own, next_below(ext)), tau);
    };

    // ---- item-factor tiles: global -> registers -> LDS, double buffered
    f32x4 st[STAGE];
    float st_b = 0.0f;
    auto stage_load = [&](int t) {
        // uniform tile base (SGPRs) + 32-bit per-thread byte offset: the scalar-base form of
        // global_load, no 64-bit VALU address arithmetic per tile
        // buffer loads: the tile's descriptor is built from scalars (base + t * tile bytes: SALU only), the thread's part
        // is one loop-invariant 32-bit offset, the quarter of the tile an SGPR offset -- no VALU address arithmetic in
        // the stretch between the barrier and the tile's first MFMA, where a dependent VALU instruction waits out one
        // MFMA of the partner wave
        char *src = const_cast<char *>(reinterpret_cast<const char *>(p.V)) + (int64_t)t * (TILE_I * KD * 4);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(src, 0, TILE_I * KD * 4, 0x00020000);
#pragma unroll
        for (int q = 0; q < STAGE; ++q) {
            typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
            const u32x4_t r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(tid * 16), q * BLOCK * 16, 0);
            st[q] = __builtin_bit_cast(f32x4, r);
        }
        if (HAS_BASE && tid < TILE_I) st_b = p.ibase[t * TILE_I + tid];
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int q = 0; q < STAGE; ++q) {
            const int idx = tid + q * BLOCK;
            const int row = idx / C4, c4 = idx - row * C4;
            *reinterpret_cast<f32x4 *>(&vt[buf][row][c4 * 4]) = st[q];
        }
    };
    if (t0 < t1) { stage_load(t0); stage_store(0); }
    __syncthreads();

    // Software pipeline: iteration t issues the MFMAs of tile t and, in the same basic block,
    // runs the branch-free select of tile t-1 (bias add, exclusion, threshold test, predicated
    // append), so the VALU/LDS work of the select fills the issue gaps of the matrix pipe.
    // Iteration t0 selects nothing (threshold +inf); iteration t1 multiplies a stale tile whose
    // result is dropped.
#ifdef IFR_STAMPS
    const bool dbg_stamp = (p.dbg & 128) && p.stamps != nullptr;       // make STAMPS=1: dev-only phase timing
#else
    constexpr bool dbg_stamp = false;
#endif
    unsigned long long ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0, ph4 = 0, ph5 = 0, ph6 = 0;
    // Two accumulators take turns (tile t multiplies into one while the select reads tile t-1 from
    // the other), so no register hand-over sits between the last MFMA of a tile and the next tile.
    // accA is what the first iteration selects from (no tile yet): -inf passes no threshold, so the select needs no
    // 'first iteration' case
    f32x16 accA, accB = {0};
#pragma unroll
    for (int r = 0; r < 16; ++r) accA[r] = -INFINITY;
    uint32_t mC = 0;
    auto tile_step = [&](int t, const f32x16 &accC, f32x16 &accN, int parity) __attribute__((always_inline)) {
        unsigned long long T0 = 0, T1 = 0, T2 = 0;
        if (dbg_stamp) T0 = __builtin_amdgcn_s_memtime();
        // dev ablations (make ABLATE=1 or STAMPS=1; otherwise compiled out: a run-time mode costs the tile loop four
        // scalar branches and a boolean carried through a VGPR): 2 = thresholds never met, 3 = no staging, no
        // barrier; 4 = no staging
#if defined(IFR_ABLATE) || defined(IFR_STAMPS)
        const int dbg_mode = p.dbg & 7;
#else
        constexpr int dbg_mode = 0;
#endif
        const bool dbg_no_stage = dbg_mode >= 3;
        const int buf = dbg_no_stage ? 0 : parity;
        // bases of tile t (st_b: loaded during the last iteration, or by the prologue) -> the slot nobody reads now
        if (HAS_BASE && tid < TILE_I) ibt[buf][tid] = st_b;
        if (WIDE && live) {
            // the user's bound word, refreshed every tile: the value loaded during the last tile raises tau now, the
            // next load is issued here and has a whole tile to come back (any workgroup of the user may have raised it)
            tau = fmaxf(tau, next_below(key_to_float(bw_seen)));
            bw_seen = __hip_atomic_load(p.bw + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // the tile's first operand read goes out before anything else: its LDS latency is the one stretch after the
        // barrier that nothing of this wave can cover
        const float *arow = &vt[buf][j][4 * h];
        f32x4 a_cur = *reinterpret_cast<const f32x4 *>(arow);
        // (likely: hipcc otherwise lays the every-tile staging and barrier out of line, behind taken branches)
        if (__builtin_expect(t + ts < t1 && !dbg_no_stage, 1)) stage_load(t + ts);

        // ---- zone A: 32 items x 32 users x KD on the matrix core (tile t), one k-group of four
        // MFMAs per step, each step followed by the branch-free part of the select for 128/KD score
        // registers of tile t-1: bias add and the threshold compare (its ballot lands in SGPRs).  sched_barrier(0) keeps the
        // steps apart so this VALU work issues while the step's last MFMA is still in the pipe.
        constexpr int G = KD / 8;               // k-groups per tile
        constexpr int SEL_G = G < 16 ? G : 16;  // k-groups that carry select work (KD > 128: the first 16)
        // score registers [16g / SEL_G, 16(g + 1) / SEL_G) are selected in k-group g (KD = 96: 1, 1, 2, 1, 1, 2, ...)
        accN = f32x16{0};
        const float seltau = dbg_mode < 2 ? tau : INFINITY;
        const int item0 = (t - ts) * TILE_I + 4 * h;          // the tile selected now: the previous one of this workgroup
        const uint32_t mm = mC >> (4 * h);
        float s[16], ibC[16];
        unsigned long long hb[16];              // per score register: ballot of the lanes whose score passes (SGPR pairs)
        __builtin_amdgcn_s_setprio(0);          // back in line for the MFMA zone (raised in zone B of the last iteration)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            f32x4 a_nxt = a_cur;
            if (g + 1 < G) a_nxt = *reinterpret_cast<const f32x4 *>(arow + 8 * (g + 1));
            if (HAS_BASE && g < 4) {
                // item bases of tile t-1: slot buf ^ 1, not written during this iteration
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(&ibt[buf ^ 1][8 * g + 4 * h]);
                ibC[4 * g + 0] = b4[0]; ibC[4 * g + 1] = b4[1]; ibC[4 * g + 2] = b4[2]; ibC[4 * g + 3] = b4[3];
            }
            if (g == 0 && dbg_stamp) { asm volatile("" ::"v"(a_cur[0])); T1 = __builtin_amdgcn_s_memtime(); }
            accN = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[0], ureg[g][0], accN, 0, 0, 0);
            accN = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[1], ureg[g][1], accN, 0, 0, 0);
            accN = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[2], ureg[g][2], accN, 0, 0, 0);
            accN = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[3], ureg[g][3], accN, 0, 0, 0);
#pragma unroll
            for (int r = g * 16 / SEL_G; r < (g < SEL_G ? (g + 1) * 16 / SEL_G : 0); ++r) {
                const float sc = HAS_BASE ? ((ibC[r] + ub) + accC[r]) : accC[r]; // (ibase + ubase) + dot
                s[r] = sc;
                hb[r] = __ballot(sc > seltau);                                  // one v_cmp into an SGPR pair
                if (STORE && (r & 3) == 3 && t > t0)                            // items item0 + 8q .. + 3 of tile t-1
                    *reinterpret_cast<f32x4 *>(drow + (int64_t)(t - ts) * TILE_I + 2 * (r - 3)) =
                        f32x4{s[r - 3], s[r - 2], s[r - 1], sc};
            }
            a_cur = a_nxt;
            __builtin_amdgcn_sched_barrier(0);
        }
        if (dbg_stamp) T2 = __builtin_amdgcn_s_memtime();

        // ---- zone B: only score registers in which some lane hits are visited, behind wave-uniform scalar
        // branches on the ballots of zone A (a hit tile used to scan every 4-register group that held a hit)
        unsigned long long T2b = 0, T2c = 0;
        // the select's VALU / LDS instructions compete with the partner wave's MFMAs for the SIMD's issue slot: while this
        // wave is outside its MFMA zone it goes first (every instruction here would otherwise wait out one more MFMA)
        __builtin_amdgcn_s_setprio(3);
        // (the union is formed here: OR-ed into one SGPR pair inside the MFMA zone, each OR waits for its v_cmp and holds
        // up the wave's next MFMA -- measured -1.9 points at C2)
        unsigned long long hany = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) hany |= hb[r];
        if (hany != 0) {
            // Straight-line append with ONE instruction on the chain from a score register to the next (the count):
            // outside its MFMA zone a wave's dependent VALU instructions each wait out an MFMA of the partner wave.
            // Every lane writes its slot min(cnt, PEND): a lane without a hit scribbles on its next free slot, a full
            // lane on the dump row pend[PEND]; only a hit advances cnt.  Exclusions (and rows past n_items) are dropped
            // here, on the rare path: the fp32 MFMA shares the VALU, so a per-score kill in zone A is paid on every tile.
            // A register without a hit costs one scalar compare + branch; nothing else is kept per register (the
            // overflow path below recomputes what it needs).  A tile without any hit takes ONE branch: not-taken
            // branches are not free either (walking all 16 on every tile cost C3 2.8 points).
            const int cnt0 = cnt;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (__builtin_expect(hb[r] != 0, 0)) {   // cold: a register without a hit falls through (no taken branch)
                    asm volatile("" ::: "memory");   // keeps this a block of its own behind a scalar branch
                    const bool hit = (s[r] > seltau) & !((mm >> ((r & 3) + 8 * (r >> 2))) & 1u);
                    pend[min(cnt, PEND)][tid] = make_uint2(__float_as_uint(s[r]), (uint32_t)(item0 + (r & 3) + 8 * (r >> 2)));
                    cnt += hit ? 1 : 0;
                }
            }
            if (dbg_stamp) T2b = __builtin_amdgcn_s_memtime();
            if (__any(cnt > PEND)) {
                // some lane ran out of slots: its hits beyond the PEND-th of this tile went to the dump row.  Find them
                // again (same walk, counting only, every register), drain every lane's slots into its register list,
                // raise tau, retry
                uint32_t todo = 0;
                int c2 = cnt0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool hit = (s[r] > seltau) & !((mm >> ((r & 3) + 8 * (r >> 2))) & 1u);
                    todo |= (hit & (c2 >= PEND)) ? (1u << r) : 0u;
                    c2 += hit ? 1 : 0;
                }
                cnt = min(cnt, PEND);
                flush(true);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if ((todo & (1u << r)) && s[r] > tau) {
                        pend[cnt][tid] = make_uint2(__float_as_uint(s[r]), (uint32_t)(item0 + (r & 3) + 8 * (r >> 2)));
                        ++cnt;
                    }
                }
            }
            if (dbg_stamp) T2c = __builtin_amdgcn_s_memtime();
        }
        // WIDE: the first tile's scores are all pending now; drain them at once, so that the list's bound word is
        // published a tile earlier (until the user's words meet, every score is a candidate)
        // ... and pending entries never wait longer than WIDE_AGE tiles: a lane's list -- hence the user's bound -- would
        // otherwise lag by up to 16 entries per list, as much as K itself, once hits have become rare
        if (WIDE) {
            ++age;
            if (t == t0 + ts || (age >= WIDE_AGE && __any(cnt > 0))) { flush(false); age = 0; }
        }
        // exclusion mask of tile t for the next iteration's select
        mC = 0;
        if (nt == t) {
            mC = nm;
            nt = nt2; nm = nm2;
            ++ex_cur;
            if (ex_cur + 1 < ex_end) { const uint2 e = p.ex_pair[ex_cur + 1]; nt2 = (int)e.x; nm2 = e.y; } else nt2 = NO_TILE;
        }
        if (WIDE) {
            // entries of tiles that other workgroups (item chunks) of this user tile sweep
            while (nt < t + ts) {
                nt = nt2; nm = nm2;
                ++ex_cur;
                if (ex_cur + 1 < ex_end) { const uint2 e = p.ex_pair[ex_cur + 1]; nt2 = (int)e.x; nm2 = e.y; } else nt2 = NO_TILE;
            }
        }
        if (t == p.n_tiles - 1) {
            const int valid = p.n_items - t * TILE_I;       // 1..32
            if (valid < 32) mC |= ~0u << valid;
        }
        if (__builtin_expect(t + ts < t1 && !dbg_no_stage, 1)) stage_store(buf ^ 1);
        if (__builtin_expect(dbg_mode != 3, 1)) __syncthreads();
        if (dbg_stamp) {
            const unsigned long long T3 = __builtin_amdgcn_s_memtime();
            ph0 += T1 - T0; ph1 += T2 - T1; ph2 += T3 - T2; ph3 += 1;
            if (T2b) { ph4 += T2b - T2; ph5 += T2c - T2b; ph6 += 1; }
        }
    };
    for (int t = t0; t <= t1; t += 2 * ts) {
        tile_step(t, accA, accB, 0);
        if (t + ts <= t1) tile_step(t + ts, accB, accA, 1);
    }
    if (dbg_stamp && lane == 0) {
        unsigned long long *o = p.stamps + ((int64_t)blockIdx.x * WAVES + wave) * 8;
        o[0] = ph0; o[1] = ph1; o[2] = ph2; o[3] = ph3; o[4] = ph4; o[5] = ph5; o[6] = ph6; o[7] = 0;
    }

    // ---- final drain and write the lane's partial list
    flush(false);
    if (live && !WIDE) {
        const int64_t base = part_off + ((slot - slot0) * lists + chunk * 2 + h) * KL;
#pragma unroll
        for (int i = 0; i < KL; ++i) { p.part_scores[base + i] = ls[i]; p.part_ids[base + i] = li[i]; }
    }
}

// ------------------------------------------------------------------------------------
// Merge the 2*n_chunks partial lists of a user into the final top-K (score desc, id asc).
// One wave per user.  Each entry's final rank = number of entries that beat it, obtained
// with one binary search per (sorted) partial list.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ bool beats(float s, int id, float s2, int id2)
{
    return (s > s2) || (s == s2 && id < id2);
}

template <int KL>
__global__ __launch_bounds__(64) void merge_topk_kernel(const float *part_scores, const int32_t *part_ids,
                                                        int64_t n, int n_lists, int stride_lists, int K,
                                                        int32_t *out_ids, float *out_scores)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *es = reinterpret_cast<float *>(smem);
    int *ei = reinterpret_cast<int *>(smem) + n_lists * KL;
    const int64_t u = blockIdx.x;
    const int lane = threadIdx.x;
    const int E = n_lists * KL;
    const float *ps = part_scores + u * (int64_t)stride_lists * KL;
    const int32_t *pi = part_ids + u * (int64_t)stride_lists * KL;
    for (int e = lane; e < E; e += 64) { es[e] = ps[e]; ei[e] = pi[e]; }
    __syncthreads();
    int n_valid = 0;
    for (int e = lane; e < E; e += 64) {
        const float s = es[e];
        const int id = ei[e];
        if (id == NO_TILE) continue;          // unfilled slot
        ++n_valid;
        int rank = 0;
        for (int l = 0; l < n_lists; ++l) {
            const float *lsp = es + l * KL;
            const int *lip = ei + l * KL;
            int lo = 0, hi = KL;                // first position that does NOT beat (s,id)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (beats(lsp[mid], lip[mid], s, id)) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        if (rank < K) { out_ids[u * K + rank] = id; out_scores[u * K + rank] = s; }
    }
    // fewer than K candidates: pad the tail
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) n_valid += __shfl_xor(n_valid, off);
    for (int r = n_valid + lane; r < K; r += 64) { out_ids[u * K + r] = -1; out_scores[u * K + r] = -INFINITY; }
}

// The same merge for short inputs (n_lists * K <= 64 * NE entries, the usual case: 8 lists x K = 10), without
// the dependent LDS reads of the binary searches: only the first K entries of a list can reach the top K, each
// becomes one 64-bit key (order-preserving score bits, then ~id, so "beats" is one unsigned compare; an unfilled
// slot is key 0), a lane keeps NE keys in registers and counts the keys that beat them in one pass over the
// wave's key array (every lane reads the same address: an LDS broadcast).  One wave per user, four users per
// workgroup.
constexpr int MERGE_WAVES = 4;

// Kl = entries read per list = min(K, KL).  K > KL (pair path): over_cnt / over_slots (not null) receive the users for
// whom a FULL list's last entry ranks before the user's K-th entry (or who have fewer than K entries in all): that
// list may have dropped an item of the top K.  slot_base = index of this launch's first user in the caller's batch.
template <int KL, int NE>
__global__ __launch_bounds__(64 * MERGE_WAVES) void merge_pairs_kernel(const float *part_scores, const int32_t *part_ids,
                                                                       int64_t n, int n_lists, int stride_lists, int K, int Kl,
                                                                       int32_t *out_ids, float *out_scores,
                                                                       uint32_t *over_cnt, int32_t *over_slots, int64_t slot_base)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t u = (int64_t)blockIdx.x * MERGE_WAVES + w;
    const int E = n_lists * Kl;                      // <= 64 * NE (host-checked)
    const int Epad = (E + 7) & ~7;
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem) + (size_t)w * Epad;
    const bool live = u < n;
    const float *ps = part_scores + (live ? u : 0) * (int64_t)stride_lists * KL;
    const int32_t *pi = part_ids + (live ? u : 0) * (int64_t)stride_lists * KL;
    unsigned long long k[NE];
    float sc[NE];
    int n_valid = 0;
#pragma unroll
    for (int q = 0; q < NE; ++q) {
        const int e = lane + 64 * q;
        k[q] = 0; sc[q] = 0.0f;
        if (e < E) {
            const int l = e / Kl, i = e - l * Kl;
            const int id = pi[l * KL + i];
            sc[q] = ps[l * KL + i];
            if (id != NO_TILE) {
                k[q] = ((unsigned long long)float_to_key(sc[q] + 0.0f) << 32) | (uint32_t)~id;   // + 0: -0 ties with +0
                ++n_valid;
            }
        }
        if (e < Epad) keys[e] = k[q];
    }
    __syncthreads();
    int rank[NE];
#pragma unroll
    for (int q = 0; q < NE; ++q) rank[q] = 0;
    for (int e = 0; e < Epad; e += 8) {
        unsigned long long x[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] = keys[e + t];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int q = 0; q < NE; ++q) rank[q] += x[t] > k[q] ? 1 : 0;
    }
    if (!live) return;
#pragma unroll
    for (int q = 0; q < NE; ++q)
        if (k[q] != 0 && rank[q] < K) { out_ids[u * K + rank[q]] = (int32_t)~(uint32_t)k[q]; out_scores[u * K + rank[q]] = sc[q]; }
    if (over_cnt != nullptr) {
        // the user's K-th key (0: fewer than K entries), then every full list's last entry against it
        unsigned long long kth = 0;
#pragma unroll
        for (int q = 0; q < NE; ++q) kth = (k[q] != 0 && rank[q] == K - 1) ? k[q] : kth;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { const unsigned long long o = __shfl_xor(kth, off); kth = o > kth ? o : kth; }
        bool lost = false;
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            const int e = lane + 64 * q;
            lost |= e < E && (e % Kl) == Kl - 1 && k[q] != 0 && k[q] > kth;
        }
        if (__any(lost) && lane == 0) over_slots[atomicAdd(over_cnt, 1u)] = (int32_t)(slot_base + u);
    }
    // fewer than K candidates: pad the tail
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) n_valid += __shfl_xor(n_valid, off);
    for (int r = n_valid + lane; r < K; r += 64) { out_ids[u * K + r] = -1; out_scores[u * K + r] = -INFINITY; }
}

// stride_lists: partial lists between two users' blocks in part_scores / part_ids (>= n_lists, the lists that are read)
// K > KL (pair path): n_lists * KL <= 256 (host-checked) and over_cnt / over_slots set; returns -1 otherwise
template <int KL>
inline int launch_merge_topk(const float *part_scores, const int32_t *part_ids, int64_t n, int n_lists, int stride_lists, int K,
                             int32_t *out_ids, float *out_scores, hipStream_t s, uint32_t *over_cnt = nullptr,
                             int32_t *over_slots = nullptr, int64_t slot_base = 0)
{
    if (n <= 0) return 0;
    const int Kl = K < KL ? K : KL;
    const int E = n_lists * Kl;
    if (K > KL && (E > 256 || over_cnt == nullptr || over_slots == nullptr)) return -1;
    if (K <= KL) { over_cnt = nullptr; over_slots = nullptr; }
    const unsigned grid4 = (unsigned)((n + MERGE_WAVES - 1) / MERGE_WAVES);
    const size_t smem4 = (size_t)MERGE_WAVES * ((E + 7) & ~7) * 8;
    if (E <= 64)
        hipLaunchKernelGGL((merge_pairs_kernel<KL, 1>), dim3(grid4), dim3(64 * MERGE_WAVES), smem4, s, part_scores, part_ids, n, n_lists, stride_lists, K, Kl, out_ids, out_scores, over_cnt, over_slots, slot_base);
    else if (E <= 128)
        hipLaunchKernelGGL((merge_pairs_kernel<KL, 2>), dim3(grid4), dim3(64 * MERGE_WAVES), smem4, s, part_scores, part_ids, n, n_lists, stride_lists, K, Kl, out_ids, out_scores, over_cnt, over_slots, slot_base);
    else if (E <= 256)
        hipLaunchKernelGGL((merge_pairs_kernel<KL, 4>), dim3(grid4), dim3(64 * MERGE_WAVES), smem4, s, part_scores, part_ids, n, n_lists, stride_lists, K, Kl, out_ids, out_scores, over_cnt, over_slots, slot_base);
    else
        hipLaunchKernelGGL((merge_topk_kernel<KL>), dim3((unsigned)n), dim3(64), (size_t)n_lists * KL * 8, s, part_scores, part_ids, n,
                           n_lists, stride_lists, K, out_ids, out_scores);
    return 0;
}

// ------------------------------------------------------------------------------------
// WIDE variant, second step: the exact top K of a user's candidate buffer (score desc, id asc).  One workgroup per
// user.  The candidates are first cut with the final bound B (the user's bound word: at least K of them reach it), which leaves about 1.3 K of them; those are ranked by counting.  A buffer that overflowed is reported
// in overflow_slots: the host ranks that user from a dense score row instead.
// ------------------------------------------------------------------------------------
constexpr int CAND_MAX = 16384;     // entries of a user's candidate buffer in global memory
constexpr int CAND_KEEP = 2048;     // ... that may reach the final bound (LDS of the selection)

struct CandParams {
    const uint2 *cand;
    const uint32_t *cand_cnt, *bw;
    int cand_cap;
    int64_t n;
    int K;
    int32_t *out_ids;
    float *out_scores;
    uint32_t *overflow_cnt;      // [1], zeroed per call
    unsigned long long *stats;   // [2], zeroed per call: candidates appended, candidates that reach the final bound (null: not counted)
    int32_t *overflow_slots;     // [n]
};

__global__ __launch_bounds__(BLOCK) void select_candidates_kernel(CandParams p)
{
    __shared__ unsigned long long keys[CAND_KEEP];
    __shared__ uint32_t sbits[CAND_KEEP];
    __shared__ int s_n;
    const int tid = threadIdx.x;
    const int64_t u = blockIdx.x;
    const uint32_t total = p.cand_cnt[u];
    const int K = p.K;
    auto overflow = [&]() {
        if (tid == 0) p.overflow_slots[atomicAdd(p.overflow_cnt, 1u)] = (int32_t)u;
        for (int r = tid; r < K; r += BLOCK) { p.out_ids[u * K + r] = -1; p.out_scores[u * K + r] = -INFINITY; }
    };
    if (total > (uint32_t)p.cand_cap) { overflow(); return; }
    const float thr = next_below(key_to_float(p.bw[u]));       // the final bound: at least K candidates reach it
    if (tid == 0) s_n = 0;
    __syncthreads();
    const uint2 *src = p.cand + u * (int64_t)p.cand_cap;
    for (uint32_t e = tid; e < total; e += BLOCK) {
        const uint2 c = src[e];
        const float x = __uint_as_float(c.x);
        if (x > thr) {
            const int pos = atomicAdd(&s_n, 1);
            if (pos < CAND_KEEP) {
                keys[pos] = ((unsigned long long)float_to_key(x + 0.0f) << 32) | (uint32_t)~c.y;     // + 0: -0 ties with +0
                sbits[pos] = c.x;
            }
        }
    }
    __syncthreads();
    const int nf = s_n;
    if (nf > CAND_KEEP) { overflow(); return; }              // uniform: the bound stayed far below the K-th best
    if (tid == 0 && p.stats != nullptr) { atomicAdd(&p.stats[0], (unsigned long long)total); atomicAdd(&p.stats[1], (unsigned long long)nf); }
    for (int e = tid; e < nf; e += BLOCK) {
        const unsigned long long k = keys[e];
        int rank = 0;
        for (int x = 0; x < nf; ++x) rank += keys[x] > k ? 1 : 0;
        if (rank < K) { p.out_ids[u * K + rank] = (int32_t)~(uint32_t)k; p.out_scores[u * K + rank] = __uint_as_float(sbits[e]); }
    }
    for (int r = nf + tid; r < K; r += BLOCK) { p.out_ids[u * K + r] = -1; p.out_scores[u * K + r] = -INFINITY; }
}

// fallback plumbing for overflowed users: user ids of the flagged slots, and their rows back into place
__global__ void gather_ids_kernel(const int32_t *user_ids, const int32_t *slots, int m, int32_t *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) out[i] = user_ids[slots[i]];
}
__global__ void scatter_rows_kernel(const int32_t *slots, int m, int K, const int32_t *ids, const float *scores, int32_t *out_ids,
                                    float *out_scores)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * K) return;
    const int r = i / K, c = i - r * K;
    out_ids[(int64_t)slots[r] * K + c] = ids[i];
    out_scores[(int64_t)slots[r] * K + c] = scores[i];
}

// ------------------------------------------------------------------------------------
// Dense score rows out[slot][item] for a batch of users (same fma chain as the fused
// kernel, operand roles swapped so a lane owns an item column and stores are coalesced).
// ------------------------------------------------------------------------------------
struct DenseParams {
    const float *U, *V, *ibase, *ubase;
    const int32_t *user_ids;     // clamped into [0, n_users) like TopkParams::user_ids
    int64_t n_users;
    int64_t n;
    int n_items, n_tiles, chunk_tiles, n_chunks;
    float *out;
    int64_t ld;
};

template <int KD, bool HAS_BASE>
__global__ __launch_bounds__(BLOCK, 2) void score_dense_kernel(DenseParams p)
{
    constexpr int ROW = KD + 4;
    constexpr int C4 = KD / 4;
    constexpr int STAGE = (TILE_I * C4) / BLOCK;
    __shared__ __attribute__((aligned(16))) float vt[2][TILE_I][ROW];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int n_ut = (int)((p.n + USERS_PER_BLOCK - 1) / USERS_PER_BLOCK);
    const int chunk = blockIdx.x / n_ut;
    const int ut = blockIdx.x - chunk * n_ut;
    const int t0 = chunk * p.chunk_tiles;
    const int t1 = min(t0 + p.chunk_tiles, p.n_tiles);

    const int64_t slot0 = (int64_t)ut * USERS_PER_BLOCK + wave * 32;
    const int64_t slot = slot0 + j;
    const int uid = (int)min((int64_t)max(p.user_ids[slot < p.n ? slot : p.n - 1], 0), p.n_users - 1);
    f32x4 ureg[KD / 8];
    {
        const float *up = p.U + (int64_t)uid * KD + 4 * h;
#pragma unroll
        for (int g = 0; g < KD / 8; ++g) ureg[g] = *reinterpret_cast<const f32x4 *>(up + 8 * g);
    }
    float ubr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t sl = slot0 + acc_row(r, h);
        ubr[r] = HAS_BASE ? p.ubase[min((int64_t)max(p.user_ids[sl < p.n ? sl : p.n - 1], 0), p.n_users - 1)] : 0.0f;
    }

    f32x4 st[STAGE];
    float st_b = 0.0f;
    auto stage_load = [&](int t) {
        // uniform tile base (SGPRs) + 32-bit per-thread byte offset: the scalar-base form of
        // global_load, no 64-bit VALU address arithmetic per tile
        char *src = const_cast<char *>(reinterpret_cast<const char *>(p.V)) + (int64_t)t * (TILE_I * KD * 4);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(src, 0, TILE_I * KD * 4, 0x00020000);
#pragma unroll
        for (int q = 0; q < STAGE; ++q) {        // scalar-addressed buffer loads, as in score_topk_kernel
            typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
            const u32x4_t r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(tid * 16), q * BLOCK * 16, 0);
            st[q] = __builtin_bit_cast(f32x4, r);
        }
        if (HAS_BASE && tid < TILE_I) st_b = p.ibase[t * TILE_I + tid];
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int q = 0; q < STAGE; ++q) {
            const int idx = tid + q * BLOCK;
            const int row = idx / C4, c4 = idx - row * C4;
            *reinterpret_cast<f32x4 *>(&vt[buf][row][c4 * 4]) = st[q];
        }
        if (HAS_BASE && tid < TILE_I) vt[buf][tid][KD] = st_b;
    };
    if (t0 < t1) { stage_load(t0); stage_store(0); }
    __syncthreads();

    // row bases of the 16 accumulator rows: one 64-bit add per store.  Rows past the batch were loaded as user n - 1
    // (factors and base above), hold that user's values and rewrite them: no per-row test in the tile loop (sixteen
    // branches per tile before)
    float *orow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t sl = slot0 + acc_row(r, h);
        orow[r] = p.out + (sl < p.n ? sl : p.n - 1) * p.ld;
    }
    // Software pipeline over two alternating accumulators: iteration t issues the 64 MFMAs of tile t, then
    // adds the bases to and stores tile t - 1 while they execute (same scheme as the fused kernel).
    f32x16 accA = {0}, accB = {0};
    float ibA = 0.0f, ibB = 0.0f;
    auto tile_step = [&](int t, f32x16 &accC, float &ibC, const f32x16 &accP, float ibP) {
        const int buf = (t - t0) & 1;
        if (t < t1) {
            if (__builtin_expect(t + 1 < t1, 1)) stage_load(t + 1);
            const float *vrow = &vt[buf][j][4 * h];
            f32x16 acc = {0};
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int g = 0; g < KD / 8; ++g) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(vrow + 8 * g);
                // A = user fragment, B = item fragment: D[user row][item col]; the products and
                // the chain order are the ones of the fused kernel (item*user, k0 then k1).
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ureg[g][0], b4[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ureg[g][1], b4[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ureg[g][2], b4[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ureg[g][3], b4[3], acc, 0, 0, 0);
            }
            accC = acc;
            ibC = HAS_BASE ? vt[buf][j][KD] : 0.0f;
        }
        __builtin_amdgcn_s_setprio(3);                   // outside the MFMA zone this wave goes first (see score_topk_kernel)
        if (t > t0) {                                    // tile t - 1
            const int item = (t - 1) * TILE_I + j;
            if (item < p.n_items) {
#pragma unroll
                for (int r = 0; r < 16; ++r) orow[r][item] = HAS_BASE ? ((ibP + ubr[r]) + accP[r]) : accP[r];
            }
        }
        if (__builtin_expect(t + 1 < t1, 1)) stage_store(buf ^ 1);
        __syncthreads();
    };
    for (int t = t0; t <= t1; t += 2) {                  // one step past the end drains the last tile
        tile_step(t, accA, ibA, accB, ibB);
        if (t + 1 <= t1) tile_step(t + 1, accB, ibB, accA, ibA);
    }
}

// ------------------------------------------------------------------------------------
// Exclusion CSR -> per-user (tile, mask) lists, written in place of the CSR row.
// ------------------------------------------------------------------------------------
__global__ void encode_exclusions_kernel(const int64_t *indptr, const int32_t *indices, int64_t n_rows,
                                         int n_items, int32_t *ex_cnt, uint2 *ex_pair)
{
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_rows) return;
    const int64_t b = indptr[u], e = indptr[u + 1];
    int tile = -1, c = 0;
    uint32_t mask = 0;
    for (int64_t i = b; i < e; ++i) {
        const int it = indices[i];
        if (it < 0 || it >= n_items) continue;
        const int T = it >> 5;
        if (T != tile) {
            if (tile >= 0) { ex_pair[b + c] = make_uint2((uint32_t)tile, mask); ++c; }
            tile = T; mask = 0;
        }
        mask |= 1u << (it & 31);
    }
    if (tile >= 0) { ex_pair[b + c] = make_uint2((uint32_t)tile, mask); ++c; }
    ex_cnt[u] = c;
}

// ------------------------------------------------------------------------------------
// Ranking metrics.  Per cut-off k six values: NDCG, Recall, Precision, HitRatio, F1, NCRR
// (cornac/metrics/ranking.py:100-123, 366-386, 329-349, 291-312, 403-431, 140-180).
// ------------------------------------------------------------------------------------
constexpr int MAX_KS = 8;
constexpr int NM = 6;                 // metric values per cut-off
#define IFR_NAN __longlong_as_double(0x7ff8000000000000LL)

struct MetricParams {
    const int32_t *user_ids;
    int64_t n;
    const int32_t *topk;         // [n][K] ranked ids (-1 = no candidate)
    int K;
    int ks[MAX_KS];
    int n_k;
    const int64_t *tr_ptr;
    const int32_t *tr_idx;
    int64_t tr_rows;
    const double *discount;      // [IFR_MAX_TOPK] 1/log2(i+2)
    double *per_user;            // [n][NM*n_k]
    int32_t *evaluated;
};

__device__ __forceinline__ bool in_sorted_row(const int32_t *idx, int64_t b, int64_t e, int id)
{
    int64_t lo = b, hi = e;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (idx[mid] < id) lo = mid + 1; else hi = mid; }
    return id >= 0 && lo < e && idx[lo] == id;
}

__device__ __forceinline__ void store_at_k(double *out, double dcg, double idcg, double crr, double icrr, int tp,
                                           int n_pos, double tp_fp)
{
    const double rc = (double)tp / (double)n_pos, pr = (double)tp / tp_fp;
    out[0] = dcg / idcg;
    out[1] = rc;
    out[2] = pr;
    out[3] = tp > 0 ? 1.0 : 0.0;
    out[4] = (pr + rc) > 0.0 ? 2.0 * (pr * rc) / (pr + rc) : 0.0;
    out[5] = tp > 0 ? crr / icrr : 0.0;
}

// one thread per user; top-K lists of at most 128 entries (the fused kernel's output)
__global__ __launch_bounds__(BLOCK) void metrics_topk_kernel(MetricParams p)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= p.n) return;
    const int ncol = NM * p.n_k;
    const int u = p.user_ids[i];
    int64_t b = 0, e = 0;
    if (u >= 0 && u < p.tr_rows) { b = p.tr_ptr[u]; e = p.tr_ptr[u + 1]; }
    const int n_pos = (int)(e - b);
    if (n_pos == 0) {                       // base_method.py:258-259: user skipped
        p.evaluated[i] = 0;
        for (int c = 0; c < ncol; ++c) p.per_user[i * ncol + c] = IFR_NAN;
        return;
    }
    p.evaluated[i] = 1;
    uint32_t hitw[4] = {0u, 0u, 0u, 0u};           // K <= 128 list positions
    int n_ranked = 0;
    for (int r = 0; r < p.K; ++r) {
        const int id = p.topk[i * (int64_t)p.K + r];
        n_ranked += id >= 0 ? 1 : 0;
        if (in_sorted_row(p.tr_idx, b, e, id)) {
            const uint32_t bit = 1u << (r & 31);
            if (r < 32) hitw[0] |= bit; else if (r < 64) hitw[1] |= bit; else if (r < 96) hitw[2] |= bit; else hitw[3] |= bit;
        }
    }
    for (int q = 0; q < p.n_k; ++q) {
        const int k = p.ks[q];
        double dcg = 0.0, idcg = 0.0, crr = 0.0, icrr = 0.0;
        int tp = 0;
        const int trunc = min(k, n_ranked);
        for (int r = 0; r < k; ++r) {
            const uint32_t hw = r < 32 ? hitw[0] : r < 64 ? hitw[1] : r < 96 ? hitw[2] : hitw[3];
            if (hw & (1u << (r & 31))) { dcg += p.discount[r]; crr += 1.0 / (double)(r + 1); ++tp; }
            if (r < n_pos) idcg += p.discount[r];
            if (r < n_pos && r < trunc) icrr += 1.0 / (double)(r + 1);
        }
        store_at_k(p.per_user + i * ncol + NM * q, dcg, idcg, crr, icrr, tp, n_pos, (double)k);
    }
}

// Full ranked lists (k = -1 or k > 32): one workgroup per user.  Extra columns after the
// NM*n_k block: MRR (NaN when nothing matches; ranking.py:213-223) and average precision for
// MAP on a tie-free ranking (ranking.py:522-525).
struct ListMetricParams {
    const int32_t *user_ids;
    int64_t n;
    const int32_t *ranked;       // [n][ld]
    int64_t ld;
    const int32_t *n_ranked;     // [n]
    int ks[MAX_KS];              // -1 = whole list
    int n_k;
    const int64_t *tr_ptr;
    const int32_t *tr_idx;
    int64_t tr_rows;
    double *per_user;            // [n][NM*n_k + 2]
    int32_t *evaluated;
};

__global__ __launch_bounds__(BLOCK) void metrics_list_kernel(ListMetricParams p)
{
    __shared__ double red[BLOCK / 64][2 * MAX_KS + 1];
    __shared__ int red_i[BLOCK / 64][MAX_KS + 2];
    __shared__ int wave_hits[BLOCK / 64];
    const int64_t i = blockIdx.x;
    const int ncol = NM * p.n_k + 2;
    const int u = p.user_ids[i];
    int64_t b = 0, e = 0;
    if (u >= 0 && u < p.tr_rows) { b = p.tr_ptr[u]; e = p.tr_ptr[u + 1]; }
    const int n_pos = (int)(e - b);
    double *out = p.per_user + i * ncol;
    if (n_pos == 0) {
        if (threadIdx.x == 0) p.evaluated[i] = 0;
        for (int c = threadIdx.x; c < ncol; c += BLOCK) out[c] = IFR_NAN;
        return;
    }
    const int len = p.n_ranked[i];
    const int32_t *row = p.ranked + i * p.ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double dcg[MAX_KS], crr[MAX_KS], ap = 0.0;
    int tp[MAX_KS], first = 0x7fffffff, seen = 0;       // seen = hits in earlier sweeps of the list
#pragma unroll
    for (int q = 0; q < MAX_KS; ++q) { dcg[q] = 0.0; crr[q] = 0.0; tp[q] = 0; }
    for (int base = 0; base < len; base += BLOCK) {
        const int pos = base + threadIdx.x;
        const bool hit = pos < len && in_sorted_row(p.tr_idx, b, e, row[pos]);
        // 1-based index of this hit among the hits of the list, in list order (for MAP)
        const unsigned long long bal = __ballot(hit);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_hits[wave] = __popcll(bal);
        __syncthreads();
        int prior = seen;
        for (int w = 0; w < wave; ++w) prior += wave_hits[w];
        int sweep = 0;
        for (int w = 0; w < BLOCK / 64; ++w) sweep += wave_hits[w];
        __syncthreads();
        seen += sweep;
        if (hit) {
            const double disc = 1.0 / log2((double)pos + 2.0);
            const double rr = 1.0 / (double)(pos + 1);
            ap += (double)(prior + before + 1) * rr;
            first = min(first, pos);
#pragma unroll
            for (int q = 0; q < MAX_KS; ++q) {
                if (q < p.n_k && (p.ks[q] <= 0 || pos < p.ks[q])) { dcg[q] += disc; crr[q] += rr; ++tp[q]; }
            }
        }
    }
    // block reduction in a fixed order
#pragma unroll
    for (int q = 0; q < MAX_KS; ++q) {
        double a = dcg[q], c = crr[q];
        int t = tp[q];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off); c += __shfl_xor(c, off); t += __shfl_xor(t, off); }
        if (lane == 0) { red[wave][2 * q] = a; red[wave][2 * q + 1] = c; red_i[wave][q] = t; }
    }
    {
        double a = ap;
        int f = first;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off); f = min(f, __shfl_xor(f, off)); }
        if (lane == 0) { red[wave][2 * MAX_KS] = a; red_i[wave][MAX_KS] = f; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        p.evaluated[i] = 1;
        double apt = 0.0;
        int ft = 0x7fffffff;
        for (int w = 0; w < BLOCK / 64; ++w) { apt += red[w][2 * MAX_KS]; ft = min(ft, red_i[w][MAX_KS]); }
        for (int q = 0; q < p.n_k; ++q) {
            double a = 0.0, c = 0.0;
            int t = 0;
            for (int w = 0; w < BLOCK / 64; ++w) { a += red[w][2 * q]; c += red[w][2 * q + 1]; t += red_i[w][q]; }
            const int k = p.ks[q];
            const int ideal = k > 0 ? min(k, n_pos) : n_pos;          // dcg_score(gt_pos, gt_pos, k)
            const int trunc = k > 0 ? min(k, len) : len;
            double idcg = 0.0, icrr = 0.0;
            for (int r = 0; r < ideal; ++r) idcg += 1.0 / log2((double)r + 2.0);
            for (int r = 0; r < min(n_pos, trunc); ++r) icrr += 1.0 / (double)(r + 1);
            store_at_k(out + NM * q, a, idcg, c, icrr, t, n_pos, k > 0 ? (double)k : (double)len);
        }
        out[NM * p.n_k] = ft == 0x7fffffff ? IFR_NAN : 1.0 / (double)(ft + 1);
        out[NM * p.n_k + 1] = apt / (double)n_pos;
    }
}

// ------------------------------------------------------------------------------------
// List metrics straight from dense score rows, without building the ranking.  Every metric of the
// list kernel above depends only on WHERE the user's positives stand in the full ranking (score
// descending, item index ascending, excluded items left out): position(p) = number of candidates that
// beat p.  One workgroup per row: the positives' keys are sorted in LDS; every item finds, by a
// branch-free search, how many positives beat IT (j) and bumps cnt[j] -- it beats the positives from
// j on --; excluded items take their bump back; a prefix sum over cnt gives every positive's position.
// O(I log P) instead of the O(I log I) sort + O(I log P) sweep; one read of the row.
// ------------------------------------------------------------------------------------
constexpr int DM_MAX_POS = 4096;
constexpr int DM_MAX_POS_F64 = 2048;     // fp64 rows: 128-bit (score key, item) keys, 20 bytes of LDS per positive

struct DenseMetricParams {
    const void *scores;          // [n][ld] fp32, or fp64 for metrics_dense_kernel<double>
    int64_t ld;
    const int32_t *user_ids;
    int n_items;                 // columns ranked (catalogue)
    const int64_t *ex_ptr;
    const int32_t *ex_idx;
    int64_t ex_rows;
    const int64_t *tr_ptr;
    const int32_t *tr_idx;
    int64_t tr_rows;
    int ks[MAX_KS];
    int n_k;
    int cap;                     // LDS entries for the positives: power of two >= the longest truth row
    double *per_user;            // [n][NM*n_k + 2]
    int32_t *evaluated;
};

__device__ __forceinline__ unsigned long long rank_key(float s, int item)
{
    uint32_t b = __float_as_uint(s);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);          // ascending-orderable
    return ((unsigned long long)(~b) << 32) | (uint32_t)item;   // smaller key = earlier in the ranking
}

// (score key, item) as ONE orderable integer, smaller = earlier in the ranking: 64 bits for fp32 rows, 128 bits for
// fp64 rows (the 64-bit image of the double, then the item), so that fp64 scores are ranked unrounded
template <typename S> struct RankKeyOf;
template <> struct RankKeyOf<float> {
    typedef unsigned long long T;
    static __device__ __forceinline__ T make(float s, int item) { return rank_key(s, item); }
};
template <> struct RankKeyOf<double> {
    typedef unsigned __int128 T;
    static __device__ __forceinline__ T make(double s, int item)
    {
        unsigned long long b = (unsigned long long)__double_as_longlong(s + 0.0);       // + 0: -0 ranks as +0
        b = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
        return ((T)(~b) << 32) | (T)(uint32_t)item;
    }
};
__device__ __forceinline__ void dm_load4(const float *p, float (&v)[4])
{
    const float4 s = *reinterpret_cast<const float4 *>(p);
    v[0] = s.x; v[1] = s.y; v[2] = s.z; v[3] = s.w;
}
__device__ __forceinline__ void dm_load4(const double *p, double (&v)[4])
{
    const double2 a = *reinterpret_cast<const double2 *>(p), b = *reinterpret_cast<const double2 *>(p + 2);
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}

template <typename S>
__global__ __launch_bounds__(BLOCK) void metrics_dense_kernel(DenseMetricParams p)
{
    typedef typename RankKeyOf<S>::T KT;
    // sized by the launch for the longest truth row of the handle (p.cap = power of two >= that): short rows
    // leave room for more resident workgroups
    extern __shared__ __attribute__((aligned(16))) unsigned char dm_lds[];
    KT *const kp = reinterpret_cast<KT *>(dm_lds);                                           // [p.cap]
    uint32_t *const cnt = reinterpret_cast<uint32_t *>(kp + p.cap);                          // [p.cap + 1]
    __shared__ double red[BLOCK / 64][2 * MAX_KS + 1];
    __shared__ int red_i[BLOCK / 64][MAX_KS + 2];
    __shared__ uint32_t wsum[BLOCK / 64];
    __shared__ int s_valid, s_nex;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i = blockIdx.x;
    const int ncol = NM * p.n_k + 2;
    const int u = p.user_ids[i];
    const S *row = reinterpret_cast<const S *>(p.scores) + i * p.ld;
    int64_t b = 0, e = 0, eb = 0, ee = 0;
    if (u >= 0 && u < p.tr_rows) { b = p.tr_ptr[u]; e = p.tr_ptr[u + 1]; }
    if (p.ex_ptr != nullptr && u >= 0 && u < p.ex_rows) { eb = p.ex_ptr[u]; ee = p.ex_ptr[u + 1]; }
    const int n_pos = (int)(e - b);
    double *out = p.per_user + i * ncol;
    if (n_pos == 0) {
        if (tid == 0) p.evaluated[i] = 0;
        for (int c = tid; c < ncol; c += BLOCK) out[c] = IFR_NAN;
        return;
    }
    if (tid == 0) { s_valid = 0; s_nex = 0; }
    __syncthreads();
    // ---- keys of the positives that are candidates, sorted ascending (best first), padded with the worst key
    int np2 = 1;
    while (np2 < n_pos) np2 <<= 1;
    {
        int v = 0;
        for (int q = tid; q < np2; q += BLOCK) {
            KT k = ~(KT)0;
            if (q < n_pos) {
                const int it = p.tr_idx[b + q];
                if (it >= 0 && it < p.n_items && !in_sorted_row(p.ex_idx, eb, ee, it)) { k = RankKeyOf<S>::make(row[it], it); ++v; }
            }
            kp[q] = k;
        }
        if (v) atomicAdd(&s_valid, v);
    }
    __syncthreads();
    for (int kk = 2; kk <= np2; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int q = tid; q < np2; q += BLOCK) {
                const int x = q ^ j;
                if (x > q) {
                    const KT a = kp[q], c = kp[x];
                    const bool up = (q & kk) == 0;
                    if ((a > c) == up) { kp[q] = c; kp[x] = a; }
                }
            }
            __syncthreads();
        }
    }
    const int P = s_valid;                                   // positives that stand in the ranking
    for (int q = tid; q <= P; q += BLOCK) cnt[q] = 0u;
    __syncthreads();
    // positives that beat `key`: branch-free lower bound over the padded array (pads beat nothing)
    auto beaten_from = [&](KT key) -> int {
        int c = 0;
        for (int step = np2 >> 1; step >= 1; step >>= 1) c += kp[c + step - 1] < key ? step : 0;
        c += kp[c] < key ? 1 : 0;
        return min(c, P);
    };
    // ---- every item bumps the first positive it beats; four independent searches per thread
    int it0 = 0;
    if ((p.ld & 3) == 0 && (reinterpret_cast<uintptr_t>(p.scores) & 15) == 0) {
        const int n4 = p.n_items & ~3;
        for (int it = 4 * tid; it < n4; it += 4 * BLOCK) {
            S sv[4];
            dm_load4(row + it, sv);
            const KT k0 = RankKeyOf<S>::make(sv[0], it), k1 = RankKeyOf<S>::make(sv[1], it + 1), k2 = RankKeyOf<S>::make(sv[2], it + 2),
                     k3 = RankKeyOf<S>::make(sv[3], it + 3);
            int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            for (int step = np2 >> 1; step >= 1; step >>= 1) {
                const KT a0 = kp[c0 + step - 1], a1 = kp[c1 + step - 1], a2 = kp[c2 + step - 1], a3 = kp[c3 + step - 1];
                c0 += a0 < k0 ? step : 0;
                c1 += a1 < k1 ? step : 0;
                c2 += a2 < k2 ? step : 0;
                c3 += a3 < k3 ? step : 0;
            }
            const KT a0 = kp[c0], a1 = kp[c1], a2 = kp[c2], a3 = kp[c3];
            c0 += a0 < k0 ? 1 : 0;
            c1 += a1 < k1 ? 1 : 0;
            c2 += a2 < k2 ? 1 : 0;
            c3 += a3 < k3 ? 1 : 0;
            if (c0 < P) atomicAdd(&cnt[c0], 1u);            // cnt[P] (beats no positive) enters no position
            if (c1 < P) atomicAdd(&cnt[c1], 1u);
            if (c2 < P) atomicAdd(&cnt[c2], 1u);
            if (c3 < P) atomicAdd(&cnt[c3], 1u);
        }
        it0 = n4;
    }
    for (int it = it0 + tid; it < p.n_items; it += BLOCK) atomicAdd(&cnt[beaten_from(RankKeyOf<S>::make(row[it], it))], 1u);
    {
        int c = 0;
        for (int64_t q = eb + tid; q < ee; q += BLOCK) {
            const int it = p.ex_idx[q];
            if (it >= 0 && it < p.n_items) { atomicSub(&cnt[beaten_from(RankKeyOf<S>::make(row[it], it))], 1u); ++c; }
        }
        if (c) atomicAdd(&s_nex, c);
    }
    __syncthreads();
    const int len = p.n_items - s_nex;                       // candidates = length of the ranking
    // ---- inclusive prefix over cnt[0..P]: cnt[q] = candidates that beat positive q, plus itself
    const int seg = (P + BLOCK) / BLOCK;                     // entries per thread, ceil((P + 1) / BLOCK)
    {
        uint32_t mine = 0;
        for (int v = 0; v < seg; ++v) { const int q = tid * seg + v; if (q <= P) mine += cnt[q]; }
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(incl, off); if (lane >= off) incl += t; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t run = incl - mine;
        for (int w = 0; w < wave; ++w) run += wsum[w];
        for (int v = 0; v < seg; ++v) { const int q = tid * seg + v; if (q <= P) { run += cnt[q]; cnt[q] = run; } }
    }
    __syncthreads();
    // ---- the metrics from the positions, hits in ranking order (q ascending)
    double dcg[MAX_KS], crr[MAX_KS], ap = 0.0;
    int tp[MAX_KS], first = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < MAX_KS; ++q) { dcg[q] = 0.0; crr[q] = 0.0; tp[q] = 0; }
    for (int q = tid; q < P; q += BLOCK) {
        const int pos = (int)cnt[q] - 1;
        const double disc = 1.0 / log2((double)pos + 2.0);
        const double rr = 1.0 / (double)(pos + 1);
        ap += (double)(q + 1) * rr;
        first = min(first, pos);
#pragma unroll
        for (int c = 0; c < MAX_KS; ++c) {
            if (c < p.n_k && (p.ks[c] <= 0 || pos < p.ks[c])) { dcg[c] += disc; crr[c] += rr; ++tp[c]; }
        }
    }
#pragma unroll
    for (int q = 0; q < MAX_KS; ++q) {
        double a = dcg[q], c = crr[q];
        int t = tp[q];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off); c += __shfl_xor(c, off); t += __shfl_xor(t, off); }
        if (lane == 0) { red[wave][2 * q] = a; red[wave][2 * q + 1] = c; red_i[wave][q] = t; }
    }
    {
        double a = ap;
        int f = first;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off); f = min(f, __shfl_xor(f, off)); }
        if (lane == 0) { red[wave][2 * MAX_KS] = a; red_i[wave][MAX_KS] = f; }
    }
    __syncthreads();
    if (tid == 0) {
        p.evaluated[i] = 1;
        double apt = 0.0;
        int ft = 0x7fffffff;
        for (int w = 0; w < BLOCK / 64; ++w) { apt += red[w][2 * MAX_KS]; ft = min(ft, red_i[w][MAX_KS]); }
        for (int q = 0; q < p.n_k; ++q) {
            double a = 0.0, c = 0.0;
            int t = 0;
            for (int w = 0; w < BLOCK / 64; ++w) { a += red[w][2 * q]; c += red[w][2 * q + 1]; t += red_i[w][q]; }
            const int k = p.ks[q];
            const int ideal = k > 0 ? min(k, n_pos) : n_pos;
            const int trunc = k > 0 ? min(k, len) : len;
            double idcg = 0.0, icrr = 0.0;
            for (int r = 0; r < ideal; ++r) idcg += 1.0 / log2((double)r + 2.0);
            for (int r = 0; r < min(n_pos, trunc); ++r) icrr += 1.0 / (double)(r + 1);
            store_at_k(out + NM * q, a, idcg, c, icrr, t, n_pos, k > 0 ? (double)k : (double)len);
        }
        out[NM * p.n_k] = ft == 0x7fffffff ? IFR_NAN : 1.0 / (double)(ft + 1);
        out[NM * p.n_k + 1] = apt / (double)n_pos;
    }
}

// Columns [n_known, n_cat) of every score row := the row's minimum over [0, n_known): items unknown to
// the model get the user's minimum score (cornac/models/recommender.py:631-635).  NaN-free rows assumed
// (fminf skips NaN like np.min does not; the reference's scores here are finite).
__global__ __launch_bounds__(BLOCK) void pad_rows_min_kernel(float *scores, int64_t ld, int n_known, int n_cat)
{
    __shared__ float red[BLOCK / 64];
    float *row = scores + (int64_t)blockIdx.x * ld;
    float m = INFINITY;
    for (int c = threadIdx.x; c < n_known; c += BLOCK) m = fminf(m, row[c]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fminf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
    for (int c = n_known + threadIdx.x; c < n_cat; c += BLOCK) row[c] = m;
}

// column sums over evaluated users, two deterministic stages (fixed assignment of rows to threads and fixed
// butterfly order, so the fp64 sums do not depend on scheduling)
constexpr int COLSUM_ROWS = 1024;
constexpr int COLSUM_BLOCK = 256;

// sum over the 256 threads of a block, identical in every thread of wave 0; `red` holds one slot per wave
__device__ __forceinline__ double block_sum_256(double v, double *red)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const double t = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(COLSUM_BLOCK) void colsum_kernel(const double *per_user, const int32_t *evaluated, int64_t n, int ncol,
                                                              double *block_sums)
{
    __shared__ double red[COLSUM_BLOCK / 64];
    const int64_t r0 = (int64_t)blockIdx.x * COLSUM_ROWS;
    for (int c = 0; c <= ncol; ++c) {                     // column ncol = the count of evaluated users
        double v = 0.0;
        for (int k = 0; k < COLSUM_ROWS / COLSUM_BLOCK; ++k) {
            const int64_t r = r0 + k * COLSUM_BLOCK + threadIdx.x;
            if (r < n && evaluated[r]) v += (c < ncol) ? per_user[r * ncol + c] : 1.0;
        }
        v = block_sum_256(v, red);
        if (threadIdx.x == 0) block_sums[(int64_t)blockIdx.x * (ncol + 1) + c] = v;
    }
}

__global__ __launch_bounds__(COLSUM_BLOCK) void reduce_sums_kernel(const double *block_sums, int n_blocks, int ncol1, double *out)
{
    __shared__ double red[COLSUM_BLOCK / 64];
    for (int c = 0; c < ncol1; ++c) {
        double v = 0.0;
        for (int b = threadIdx.x; b < n_blocks; b += COLSUM_BLOCK) v += block_sums[(int64_t)b * ncol1 + c];
        v = block_sum_256(v, red);
        if (threadIdx.x == 0) out[c] = v;
    }
}

}  // namespace ifr
