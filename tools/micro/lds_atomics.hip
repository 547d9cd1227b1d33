// Dev microbenchmark: LDS atomic throughput on gfx950 (random bank-spread addresses).
// hipcc --offload-arch=gfx950 -O3 tools/micro/lds_atomics.hip -o tools/micro/lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int N = 4096, ITERS = 2000;
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long *out, int seed)
{
    __shared__ unsigned long long a64[N];
    uint32_t *a32 = reinterpret_cast<uint32_t *>(a64);
    for (int i = threadIdx.x; i < N; i += 256) a64[i] = 0;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + seed + blockIdx.x;
    unsigned long long acc = 0;
    for (int it = 0; it < ITERS; ++it) {
        x = x * 1664525u + 1013904223u;
        const int s = (x >> 12) & (N - 1);
        if (MODE == 0) atomicAdd(&a32[s], 1u);
        if (MODE == 1) acc += atomicAdd(&a32[s], 1u);
        if (MODE == 2) acc += atomicCAS(&a32[s], 0u, x | 1u);
        if (MODE == 3) atomicAdd(&a64[s], (unsigned long long)x);
        if (MODE == 4) acc += atomicAdd(&a64[s], (unsigned long long)x);
        if (MODE == 5) acc += atomicOr(&a32[s], 1u << (x & 31));
        if (MODE == 6) a32[s] = x;                       // plain store
        if (MODE == 7) acc += a32[s];                    // plain load
        if (MODE == 8) { a64[s] = x; }                   // 64-bit store
        if (MODE == 9) { atomicAdd(&a32[2 * s], x); atomicAdd(&a32[2 * s + 1], 1u); }   // two 32-bit adds (no carry)
    }
    __syncthreads();
    if (acc == 0x12345) out[0] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = a64[5];
}
template <int MODE>
void run(const char *name, unsigned long long *d, int blocks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, 2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // blocks spread over 256 CUs; per-CU lane-ops
    const double ops_per_cu = (double)blocks / 256.0 * 256.0 * ITERS;
    printf("%-28s blocks %5d: %8.3f ms  %6.2f lane-ops/clk/CU (2.4 GHz)\n", name, blocks, ms, ops_per_cu / (ms * 1e-3 * 2.4e9));
}
int main()
{
    unsigned long long *d;
    hipMalloc(&d, 64);
    for (int blocks : {256, 768, 2048}) {
        run<0>("add32", d, blocks);
        run<1>("add32 rtn", d, blocks);
        run<2>("cas32 rtn", d, blocks);
        run<3>("add64", d, blocks);
        run<4>("add64 rtn", d, blocks);
        run<5>("or32 rtn", d, blocks);
        run<6>("store32", d, blocks);
        run<7>("load32", d, blocks);
        run<8>("store64", d, blocks);
        run<9>("2 x add32", d, blocks);
    }
    return 0;
}
