// Micro-benchmarks that price the building blocks of the count sampler on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../prosstt_amd/csrc/prnb_device.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void bench(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = tid ^ seed;
    float facc = (float)(tid & 1023) * 1e-3f + 0.5f;
    __shared__ float inv_k[prnb::kKTab];
    for (int k = threadIdx.x; k < prnb::kKTab; k += 256) inv_k[k] = k ? 1.0f / (float)k : 0.0f;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {            // Philox4x32-10
            prnb::Words w = prnb::philox4x32_10(tid, i, acc, 0u, seed, 17u);
            acc ^= w.w[0] ^ w.w[1] ^ w.w[2] ^ w.w[3];
        } else if (MODE == 1) {     // 40 dependent v_mul_lo_u32
#pragma unroll
            for (int j = 0; j < 40; ++j) acc = acc * 0xD2511F53u + 1u;
        } else if (MODE == 2) {     // 40 dependent v_fma_f32
#pragma unroll
            for (int j = 0; j < 40; ++j) facc = __builtin_fmaf(facc, 0.999f, 0.001f);
        } else if (MODE == 3) {     // 40 v_mul_hi_u32
#pragma unroll
            for (int j = 0; j < 40; ++j) acc = __umulhi(acc, 0xCD9E8D57u) + 0x9E3779B9u;
        } else if (MODE == 4) {     // light setup: params + log1p + exp
            prnb::Params P = prnb::make_params(facc, 1.0f + (float)(i & 7), 0.2f, 1.3f);
            const float t = P.m * (prnb::det_log1p(P.theta) * P.inv_th);
            facc = prnb::det_exp(-t) + 0.5f;
        } else if (MODE == 5) {     // hardware transcendentals for comparison
            float th = __builtin_fmaf(0.2f, facc, 1.3f);
            float t = facc * (__logf(1.0f + th) * __builtin_amdgcn_rcpf(th));
            facc = __expf(-t) + 0.5f;
        } else if (MODE == 6) {     // full light draw, uniform parameters across the wave (no divergence in params)
            prnb::Params P = prnb::make_params(1.5f, 1.0f + 0.01f * (float)(i & 7), 0.2f, 1.3f);
            prnb::Words w = prnb::philox4x32_10(tid, i, 0u, 0u, seed, 17u);
            acc += prnb::light_draw(P, w.w[0], inv_k) + prnb::light_draw(P, w.w[1], inv_k) +
                   prnb::light_draw(P, w.w[2], inv_k) + prnb::light_draw(P, w.w[3], inv_k);
        } else if (MODE == 7) {     // 24-bit multiply
#pragma unroll
            for (int j = 0; j < 40; ++j) acc = __umul24(acc, 0x511F53u) + 1u;
        }
    }
    out[tid] = acc + __float_as_uint(facc);
}

template <int MODE>
static int run(const char* name, double units_per_iter)
{
    const int blocks = 256 * 8, iters = 256;
    uint32_t* d;
    CK(hipMalloc(&d, blocks * 256 * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    bench<MODE><<<blocks, 256>>>(d, iters, 1);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    bench<MODE><<<blocks, 256>>>(d, iters, 2);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    double n = (double)blocks * 256 * iters * units_per_iter;
    printf("%-28s %8.3f ms  %10.2f G units/s   (%.1f lane-cycles/unit at 32768 lanes x 2.4 GHz)\n", name, ms,
           n / ms / 1e6, 32768.0 * 2.4e9 * ms * 1e-3 / n);
    CK(hipFree(d));
    return 0;
}

int main()
{
    run<0>("philox4x32-10 (per call)", 1);
    run<1>("v_mul_lo_u32+add (per op)", 40);
    run<2>("v_fma_f32 (per op)", 40);
    run<3>("v_mul_hi_u32+add (per op)", 40);
    run<7>("v_mul_u32_u24+add (per op)", 40);
    run<4>("det setup (params,log1p,exp)", 1);
    run<5>("hw  setup (log,rcp,exp)", 1);
    run<6>("light draw m=1.5 (per 4)", 1);
    return 0;
}
