// packed vs scalar binary32 FMA issue rate on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int MODE>
__global__ __launch_bounds__(256) void bench(float* out, int iters)
{
    float t = threadIdx.x * 1e-3f;
    f2 a = {t, t + 1.f}, b = {t + 2.f, t + 3.f}, c = {t + .5f, t + .25f}, d = {t + .125f, t + .75f};
    float s0 = t, s1 = t + 1.f, s2 = t + 2.f, s3 = t + 3.f;
    const f2 m = {0.999f, 0.998f}, k = {0.001f, 0.002f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) {   // 4 independent scalar fma chains
                s0 = __builtin_fmaf(s0, 0.999f, 0.001f); s1 = __builtin_fmaf(s1, 0.998f, 0.002f);
                s2 = __builtin_fmaf(s2, 0.997f, 0.003f); s3 = __builtin_fmaf(s3, 0.996f, 0.004f);
            } else {           // 4 independent packed fma chains (8 floats)
                a = __builtin_elementwise_fma(a, m, k); b = __builtin_elementwise_fma(b, m, k);
                c = __builtin_elementwise_fma(c, m, k); d = __builtin_elementwise_fma(d, m, k);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = s0 + s1 + s2 + s3 + a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
}
template <int MODE> int run(const char* name, double flo)
{
    const int blocks = 256 * 8, iters = 512;
    float* d; CK(hipMalloc(&d, blocks * 256 * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    bench<MODE><<<blocks, 256>>>(d, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); bench<MODE><<<blocks, 256>>>(d, iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double n = (double)blocks * 256 * iters * 16 * 4;
    printf("%-22s %7.3f ms  %8.1f G instr-lanes/s  %8.1f G float-fma/s\n", name, ms, n / ms / 1e6, n * flo / ms / 1e6);
    return 0;
}
int main() { run<0>("v_fma_f32", 1); run<1>("v_pk_fma_f32", 2); return 0; }
