// Philox4x32-10 vs Threefry4x32-{20,12} issue cost on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../prosstt_amd/csrc/prnb_device.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int ROUNDS>
__device__ __forceinline__ prnb::Words threefry4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                    uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3)
{
    const uint32_t ks[5] = {k0, k1, k2, k3, 0x1BD11BDAu ^ k0 ^ k1 ^ k2 ^ k3};
    uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1], x2 = c2 + ks[2], x3 = c3 + ks[3];
    constexpr int R[8][2] = {{10, 26}, {11, 21}, {13, 27}, {23, 5}, {6, 20}, {17, 11}, {25, 10}, {18, 20}};
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int ra = R[r & 7][0], rb = R[r & 7][1];
        if ((r & 1) == 0) {
            x0 += x1; x1 = (x1 << ra) | (x1 >> (32 - ra)); x1 ^= x0;
            x2 += x3; x3 = (x3 << rb) | (x3 >> (32 - rb)); x3 ^= x2;
        } else {
            x0 += x3; x3 = (x3 << ra) | (x3 >> (32 - ra)); x3 ^= x0;
            x2 += x1; x1 = (x1 << rb) | (x1 >> (32 - rb)); x1 ^= x2;
        }
        if ((r & 3) == 3) {
            const int s = r / 4 + 1;
            x0 += ks[s % 5]; x1 += ks[(s + 1) % 5]; x2 += ks[(s + 2) % 5]; x3 += ks[(s + 3) % 5] + (uint32_t)s;
        }
    }
    prnb::Words w; w.w[0] = x0; w.w[1] = x1; w.w[2] = x2; w.w[3] = x3; return w;
}

template <int MODE>
__global__ __launch_bounds__(256) void bench(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, acc = tid ^ seed;
    for (int i = 0; i < iters; ++i) {
        prnb::Words w;
        if (MODE == 0) w = prnb::philox4x32_10(tid, i, acc, 0u, seed, 17u);
        if (MODE == 1) w = threefry4x32<20>(tid, i, acc, 0u, seed, 17u, 0u, 0u);
        if (MODE == 2) w = threefry4x32<12>(tid, i, acc, 0u, seed, 17u, 0u, 0u);
        acc ^= w.w[0] ^ w.w[1] ^ w.w[2] ^ w.w[3];
    }
    out[tid] = acc;
}
__global__ void kat(uint32_t* out)
{
    prnb::Words a = threefry4x32<20>(0, 0, 0, 0, 0, 0, 0, 0);
    prnb::Words b = threefry4x32<20>(~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u);
    prnb::Words c = threefry4x32<20>(0x243f6a88u, 0x85a308d3u, 0x13198a2eu, 0x03707344u, 0xa4093822u, 0x299f31d0u, 0x082efa98u, 0xec4e6c89u);
    for (int i = 0; i < 4; ++i) { out[i] = a.w[i]; out[4 + i] = b.w[i]; out[8 + i] = c.w[i]; }
}
template <int MODE> int run(const char* name)
{
    const int blocks = 256 * 8, iters = 256;
    uint32_t* d; CK(hipMalloc(&d, blocks * 256 * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    bench<MODE><<<blocks, 256>>>(d, iters, 1); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); bench<MODE><<<blocks, 256>>>(d, iters, 2); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-20s %7.3f ms  %8.1f G calls/s\n", name, ms, (double)blocks * 256 * iters / ms / 1e6);
    return 0;
}
int main()
{
    run<0>("philox4x32-10"); run<1>("threefry4x32-20"); run<2>("threefry4x32-12");
    uint32_t* d; CK(hipMalloc(&d, 64)); kat<<<1, 1>>>(d); uint32_t h[12]; CK(hipMemcpy(h, d, 48, hipMemcpyDeviceToHost));
    for (int i = 0; i < 12; ++i) printf("%08x%s", h[i], (i % 4 == 3) ? "\n" : " ");
    return 0;
}
