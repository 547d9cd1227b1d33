// Dev microbenchmark: cost of an LDS atomic wave instruction on gfx950 as a function of its active lanes.
// hipcc --offload-arch=gfx950 -O3 tools/micro/lds_sparse_lanes.hip -o tools/micro/lds_sparse_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int N = 4096, ITERS = 2000;
template <int MODE, int ACT>
__global__ __launch_bounds__(512) void k(unsigned long long *out, int seed)
{
    __shared__ __attribute__((aligned(16))) unsigned long long a64[N];
    uint32_t *a32 = reinterpret_cast<uint32_t *>(a64);
    for (int i = threadIdx.x; i < N; i += 512) a64[i] = 0;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + seed + blockIdx.x;
    unsigned long long acc = 0;
    const bool on = (int)(threadIdx.x & 63) < ACT;
    for (int it = 0; it < ITERS; ++it) {
        x = x * 1664525u + 1013904223u;
        const int s = (x >> 12) & (N - 1);
        if (on) {
            if (MODE == 0) acc += atomicCAS(&a32[s], 0u, x | 1u);
            if (MODE == 1) atomicAdd(&a64[s], (unsigned long long)x);
            if (MODE == 2) { typedef uint32_t u4 __attribute__((ext_vector_type(4))); const u4 v = *reinterpret_cast<const u4 *>(&a32[(s & ~3)]); acc += v[0] ^ v[1] ^ v[2] ^ v[3]; }
            if (MODE == 3) acc += a32[s];
        }
    }
    __syncthreads();
    if (acc == 0x12345) out[0] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = a64[5];
}
template <int MODE, int ACT>
void run(const char *name, unsigned long long *d)
{
    const int blocks = 768;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, ACT><<<blocks, 512>>>(d, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, ACT><<<blocks, 512>>>(d, 2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double winstr_per_cu = (double)blocks / 256.0 * 8.0 * ITERS;     // wave instructions per CU
    printf("%-12s active %2d: %8.3f ms  %7.2f clk per wave instruction per CU (2.4 GHz)\n", name, ACT, ms, ms * 1e-3 * 2.4e9 / winstr_per_cu);
}
int main()
{
    unsigned long long *d;
    hipMalloc(&d, 64);
    run<0, 64>("cas32 rtn", d); run<0, 16>("cas32 rtn", d); run<0, 4>("cas32 rtn", d); run<0, 1>("cas32 rtn", d);
    run<1, 64>("add64", d); run<1, 16>("add64", d); run<1, 4>("add64", d); run<1, 1>("add64", d);
    run<2, 64>("read b128", d); run<2, 16>("read b128", d); run<2, 4>("read b128", d);
    run<3, 64>("read b32", d); run<3, 16>("read b32", d); run<3, 4>("read b32", d);
    return 0;
}
