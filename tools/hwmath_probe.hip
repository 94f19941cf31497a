// Accuracy of gfx950's v_rcp_f32 / v_log_f32 / v_exp_f32 against binary64 over the argument ranges the
// K3 fast path uses.  hipcc --offload-arch=gfx950 -O3 -o hwmath_probe hwmath_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdint>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// mode 0: rcp(x); 1: log2(x); 2: exp2(-x); 3: f(theta) = log1p(theta)/theta as log2(u1)/(u1-1)*ln2 with u1 = 1+theta
__global__ void probe(int mode, uint32_t lo_bits, uint32_t n, double* maxerr, double* sumerr)
{
    double worst = 0.0, sum = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float(lo_bits + (uint32_t)i);
        double got, ref;
        if (mode == 0) { got = __builtin_amdgcn_rcpf(x); ref = 1.0 / (double)x; }
        else if (mode == 1) { got = __builtin_amdgcn_logf(x); ref = log2((double)x); }
        else if (mode == 2) { got = __builtin_amdgcn_exp2f(-x); ref = exp2(-(double)x); }
        else {
            const float u1 = 1.0f + x, d = u1 - 1.0f;
            got = (double)(__builtin_amdgcn_logf(u1) * __builtin_amdgcn_rcpf(d)) * 0.6931471805599453;
            ref = log1p((double)x) / (double)x;
        }
        const double rel = fabs(got - ref) / fabs(ref);
        if (ref != 0.0 && rel == rel) { if (rel > worst) worst = rel; sum += rel; }
    }
    // block reduce via atomics on ordered doubles (positive): use unsigned long long max
    atomicMax((unsigned long long*)maxerr, (unsigned long long)__double_as_longlong(worst));
    atomicAdd(sumerr, sum);
}

static int run(const char* name, int mode, float lo, float hi)
{
    uint32_t lb, hb; memcpy(&lb, &lo, 4); memcpy(&hb, &hi, 4);
    const uint32_t n = hb - lb;
    double *d_max, *d_sum; CK(hipMalloc(&d_max, 8)); CK(hipMalloc(&d_sum, 8));
    CK(hipMemset(d_max, 0, 8)); CK(hipMemset(d_sum, 0, 8));
    probe<<<4096, 256>>>(mode, lb, n, d_max, d_sum);
    CK(hipDeviceSynchronize());
    double m, s; CK(hipMemcpy(&m, d_max, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&s, d_sum, 8, hipMemcpyDeviceToHost));
    printf("%-34s [%g, %g): %u values, max rel err %.3e (2^%.2f), mean %.3e\n", name, lo, hi, n, m, log2(m), s / n);
    return 0;
}

int main()
{
    run("v_rcp_f32", 0, 1.1920929e-7f, 64.0f);
    run("v_log_f32 (log2)", 1, 1.0000001f, 1.001f);
    run("v_log_f32 (log2)", 1, 1.001f, 1.25f);
    run("v_log_f32 (log2)", 1, 1.25f, 32.0f);
    run("v_exp_f32 (2^-x)", 2, 1e-6f, 1.0f);
    run("v_exp_f32 (2^-x)", 2, 1.0f, 32.0f);
    run("f(theta)=log2(1+th)/(1+th-1)*ln2", 3, 1e-7f, 1e-3f);
    run("f(theta)", 3, 1e-3f, 0.25f);
    run("f(theta)", 3, 0.25f, 16.0f);
    return 0;
}
