// Issue cost of the instruction kinds the K3 stream kernel is made of, on gfx950:
// cycles per wave-instruction per SIMD at 4 waves per SIMD, for pure streams and for
// VALU/SALU mixes inside one wave.  hipcc --offload-arch=gfx950 -O3 -o microbench4 microbench4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP8(X) X X X X X X X X

// one "group" = 8 instructions of the kind under test (for mixes: 8 VALU + n SALU)
template <int MODE>
__global__ __launch_bounds__(1024) void bench(uint32_t* out, int iters, uint32_t seed, unsigned long long* clk)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    float a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3, a4 = tid + 4, a5 = tid + 5, a6 = tid + 6, a7 = tid + 7;
    float x = 1.0000001f, y = 1e-9f;
    uint32_t u0 = tid, u1 = tid * 3, u2 = tid * 5, u3 = tid * 7, u4 = tid ^ 9, u5 = tid ^ 11, u6 = tid ^ 13, u7 = tid ^ 15;
    uint32_t s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3, s4 = seed + 4, s5 = seed + 5, s6 = seed + 6, s7 = seed + 7;
    unsigned long long m0 = seed, m1 = seed * 3ull, m2 = 5, m3 = 7;
    uint64_t w0 = tid, w1 = tid + 1, w2 = tid + 2, w3 = tid + 3;
    __shared__ uint4 lds[1024];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 q4 = {u0, u1, u2, u3};
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 d0 = {a0, a1}, d1 = {a2, a3}, d2 = {a4, a5}, d3 = {a6, a7}, d4 = {x, x};
    const unsigned long long t0 = clock64();
    const unsigned long long r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {   // v_fma_f32
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
        } else if (MODE == 1) {   // v_xor_b32
            REP8(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                              "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(seed));)
        } else if (MODE == 2) {   // v_mad_u64_u32
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, 0\n v_mad_u64_u32 %1, vcc, %4, %5, 0\n v_mad_u64_u32 %2, vcc, %4, %5, 0\n v_mad_u64_u32 %3, vcc, %4, %5, 0\n"
                              "v_mad_u64_u32 %0, vcc, %4, %5, 0\n v_mad_u64_u32 %1, vcc, %4, %5, 0\n v_mad_u64_u32 %2, vcc, %4, %5, 0\n v_mad_u64_u32 %3, vcc, %4, %5, 0\n"
                              : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(u0), "v"(seed) : "vcc");)
        } else if (MODE == 3) {   // s_add_u32
            REP8(asm volatile("s_add_u32 %0, %0, %8\n s_add_u32 %1, %1, %8\n s_add_u32 %2, %2, %8\n s_add_u32 %3, %3, %8\n"
                              "s_add_u32 %4, %4, %8\n s_add_u32 %5, %5, %8\n s_add_u32 %6, %6, %8\n s_add_u32 %7, %7, %8\n"
                              : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "s"(seed) : "scc");)
        } else if (MODE == 4) {   // 8 v_fma + 8 s_add interleaved
            REP8(asm volatile("v_fma_f32 %0, %0, %16, %17\n s_add_u32 %8, %8, %18\n v_fma_f32 %1, %1, %16, %17\n s_add_u32 %9, %9, %18\n"
                              "v_fma_f32 %2, %2, %16, %17\n s_add_u32 %10, %10, %18\n v_fma_f32 %3, %3, %16, %17\n s_add_u32 %11, %11, %18\n"
                              "v_fma_f32 %4, %4, %16, %17\n s_add_u32 %12, %12, %18\n v_fma_f32 %5, %5, %16, %17\n s_add_u32 %13, %13, %18\n"
                              "v_fma_f32 %6, %6, %16, %17\n s_add_u32 %14, %14, %18\n v_fma_f32 %7, %7, %16, %17\n s_add_u32 %15, %15, %18\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                                "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7)
                              : "v"(x), "v"(y), "s"(seed) : "scc");)
        } else if (MODE == 5) {   // 8 v_fma + 4 s_and_b64
            REP8(asm volatile("v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %1, %1, %12, %13\n s_and_b64 %8, %8, %9\n"
                              "v_fma_f32 %2, %2, %12, %13\n v_fma_f32 %3, %3, %12, %13\n s_and_b64 %9, %9, %10\n"
                              "v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %12, %13\n s_and_b64 %10, %10, %11\n"
                              "v_fma_f32 %6, %6, %12, %13\n v_fma_f32 %7, %7, %12, %13\n s_and_b64 %11, %11, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                                "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : "v"(x), "v"(y) : "scc");)
        } else if (MODE == 6) {   // v_cmp_lt_f32 -> sgpr pair (VOP3)
            REP8(asm volatile("v_cmp_lt_f32 %0, %4, %5\n v_cmp_lt_f32 %1, %5, %6\n v_cmp_lt_f32 %2, %6, %7\n v_cmp_lt_f32 %3, %7, %4\n"
                              "v_cmp_lt_f32 %0, %4, %6\n v_cmp_lt_f32 %1, %5, %7\n v_cmp_lt_f32 %2, %6, %4\n v_cmp_lt_f32 %3, %7, %5\n"
                              : "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
        } else if (MODE == 7) {   // v_cndmask_b32 with sgpr mask
            REP8(asm volatile("v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n"
                              "v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9\n"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(seed), "s"(m0));)
        } else if (MODE == 8) {   // v_cvt_f32_u32
            REP8(asm volatile("v_cvt_f32_u32 %0, %8\n v_cvt_f32_u32 %1, %9\n v_cvt_f32_u32 %2, %10\n v_cvt_f32_u32 %3, %11\n"
                              "v_cvt_f32_u32 %4, %8\n v_cvt_f32_u32 %5, %9\n v_cvt_f32_u32 %6, %10\n v_cvt_f32_u32 %7, %11\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));)
        } else if (MODE == 9) {   // v_mbcnt_lo + v_mbcnt_hi
            REP8(asm volatile("v_mbcnt_lo_u32_b32 %0, %4, 0\n v_mbcnt_hi_u32_b32 %0, %5, %0\n v_mbcnt_lo_u32_b32 %1, %4, 0\n v_mbcnt_hi_u32_b32 %1, %5, %1\n"
                              "v_mbcnt_lo_u32_b32 %2, %4, 0\n v_mbcnt_hi_u32_b32 %2, %5, %2\n v_mbcnt_lo_u32_b32 %3, %4, 0\n v_mbcnt_hi_u32_b32 %3, %5, %3\n"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "s"(s0), "s"(s1));)
        } else if (MODE == 10) {  // v_mul_f32 (VOP2)
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                              "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));)
        } else if (MODE == 11) {  // 8 v_fma + 16 s_add
            REP8(asm volatile("v_fma_f32 %0, %0, %16, %17\n s_add_u32 %8, %8, %18\n s_add_u32 %9, %9, %18\n v_fma_f32 %1, %1, %16, %17\n s_add_u32 %10, %10, %18\n s_add_u32 %11, %11, %18\n"
                              "v_fma_f32 %2, %2, %16, %17\n s_add_u32 %12, %12, %18\n s_add_u32 %13, %13, %18\n v_fma_f32 %3, %3, %16, %17\n s_add_u32 %14, %14, %18\n s_add_u32 %15, %15, %18\n"
                              "v_fma_f32 %4, %4, %16, %17\n s_add_u32 %8, %8, %18\n s_add_u32 %9, %9, %18\n v_fma_f32 %5, %5, %16, %17\n s_add_u32 %10, %10, %18\n s_add_u32 %11, %11, %18\n"
                              "v_fma_f32 %6, %6, %16, %17\n s_add_u32 %12, %12, %18\n s_add_u32 %13, %13, %18\n v_fma_f32 %7, %7, %16, %17\n s_add_u32 %14, %14, %18\n s_add_u32 %15, %15, %18\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                                "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7)
                              : "v"(x), "v"(y), "s"(seed) : "scc");)
        } else if (MODE == 12) {  // ds_write_b128, all lanes, distinct slots
            REP8(asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n ds_write_b128 %0, %1 offset:2048\n ds_write_b128 %0, %1 offset:3072\n"
                              "ds_write_b128 %0, %1 offset:4096\n ds_write_b128 %0, %1 offset:5120\n ds_write_b128 %0, %1 offset:6144\n ds_write_b128 %0, %1 offset:7168\n"
                              "s_waitcnt lgkmcnt(0)\n"
                              :: "v"((threadIdx.x & 63) * 16), "v"(q4) : "memory");)
        } else if (MODE == 20) {  // v_fmac_f32 (VOP2)
            REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                              "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
        } else if (MODE == 21) {  // v_fmaak_f32 (VOP2 + literal)
            REP8(asm volatile("v_fmaak_f32 %0, %0, %8, 0x3f800001\n v_fmaak_f32 %1, %1, %8, 0x3f800001\n v_fmaak_f32 %2, %2, %8, 0x3f800001\n v_fmaak_f32 %3, %3, %8, 0x3f800001\n"
                              "v_fmaak_f32 %4, %4, %8, 0x3f800001\n v_fmaak_f32 %5, %5, %8, 0x3f800001\n v_fmaak_f32 %6, %6, %8, 0x3f800001\n v_fmaak_f32 %7, %7, %8, 0x3f800001\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));)
        } else if (MODE == 22) {  // v_cmp_lt_f32 e32 -> vcc
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0\n"
                              "v_cmp_lt_f32 vcc, %0, %2\n v_cmp_lt_f32 vcc, %1, %3\n v_cmp_lt_f32 vcc, %2, %0\n v_cmp_lt_f32 vcc, %3, %1\n"
                              :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");)
        } else if (MODE == 23) {  // v_cndmask_b32 e32 (vcc)
            REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                              "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(seed));)
        } else if (MODE == 24) {  // v_add_u32 (VOP2)
            REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                              "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(seed));)
        } else if (MODE == 25) {  // v_lshl_add_u32 (VOP3)
            REP8(asm volatile("v_lshl_add_u32 %0, %0, 4, %8\n v_lshl_add_u32 %1, %1, 4, %8\n v_lshl_add_u32 %2, %2, 4, %8\n v_lshl_add_u32 %3, %3, 4, %8\n"
                              "v_lshl_add_u32 %4, %4, 4, %8\n v_lshl_add_u32 %5, %5, 4, %8\n v_lshl_add_u32 %6, %6, 4, %8\n v_lshl_add_u32 %7, %7, 4, %8\n"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(seed));)
        } else if (MODE == 26) {  // v_xor_b32 with sgpr source
            REP8(asm volatile("v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n"
                              "v_xor_b32 %4, %8, %4\n v_xor_b32 %5, %8, %5\n v_xor_b32 %6, %8, %6\n v_xor_b32 %7, %8, %7\n"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "s"(seed));)
        } else if (MODE == 27) {  // v_mad_u64_u32 with sgpr multiplier (the Philox form)
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, 0\n v_mad_u64_u32 %1, vcc, %4, %5, 0\n v_mad_u64_u32 %2, vcc, %4, %5, 0\n v_mad_u64_u32 %3, vcc, %4, %5, 0\n"
                              "v_mad_u64_u32 %0, vcc, %4, %5, 0\n v_mad_u64_u32 %1, vcc, %4, %5, 0\n v_mad_u64_u32 %2, vcc, %4, %5, 0\n v_mad_u64_u32 %3, vcc, %4, %5, 0\n"
                              : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(u0), "s"(seed) : "vcc");)
        } else if (MODE == 28) {  // v_mul_hi_u32 / v_mul_lo_u32 pairs
            REP8(asm volatile("v_mul_hi_u32 %0, %4, %5\n v_mul_lo_u32 %1, %4, %5\n v_mul_hi_u32 %2, %4, %5\n v_mul_lo_u32 %3, %4, %5\n"
                              "v_mul_hi_u32 %0, %4, %5\n v_mul_lo_u32 %1, %4, %5\n v_mul_hi_u32 %2, %4, %5\n v_mul_lo_u32 %3, %4, %5\n"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u4), "v"(seed));)
        } else if (MODE == 29) {  // v_pk_mul_f32
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                              "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d4));)
        } else if (MODE == 30) {  // v_addc_co_u32 f = f + f + mask bit
            REP8(asm volatile("v_addc_co_u32 %0, vcc, %0, %0, %8\n v_addc_co_u32 %1, vcc, %1, %1, %8\n v_addc_co_u32 %2, vcc, %2, %2, %8\n v_addc_co_u32 %3, vcc, %3, %3, %8\n"
                              "v_addc_co_u32 %4, vcc, %4, %4, %8\n v_addc_co_u32 %5, vcc, %5, %5, %8\n v_addc_co_u32 %6, vcc, %6, %6, %8\n v_addc_co_u32 %7, vcc, %7, %7, %8\n"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "s"(m0) : "vcc");)
        } else if (MODE == 31) {  // v_max3_f32
            REP8(asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                              "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
        } else if (MODE == 32) {  // ds_write_b64 x8 + wait
            REP8(asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:512\n ds_write_b64 %0, %1 offset:1024\n ds_write_b64 %0, %1 offset:1536\n"
                              "ds_write_b64 %0, %1 offset:2048\n ds_write_b64 %0, %1 offset:2560\n ds_write_b64 %0, %1 offset:3072\n ds_write_b64 %0, %1 offset:3584\n"
                              "s_waitcnt lgkmcnt(0)\n"
                              :: "v"((threadIdx.x & 63) * 8), "v"(w0) : "memory");)
        } else if (MODE == 33) {  // 8 v_fma + 8 ds_write_b128 (do LDS stores hide under VALU?)
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n ds_write_b128 %10, %11\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n ds_write_b128 %10, %11 offset:1024\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n ds_write_b128 %10, %11 offset:2048\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n ds_write_b128 %10, %11 offset:3072\n v_fma_f32 %7, %7, %8, %9\n"
                              "s_waitcnt lgkmcnt(0)\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y), "v"((threadIdx.x & 63) * 16), "v"(q4) : "memory");)
        } else if (MODE == 13) {  // v_cmp to vcc + s_and_saveexec + s_or exec (the push idiom without the store)
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_and_saveexec_b64 %2, vcc\n v_fma_f32 %0, %0, %4, %5\n s_or_b64 exec, exec, %2\n"
                              "v_cmp_lt_f32 vcc, %1, %0\n s_and_saveexec_b64 %3, vcc\n v_fma_f32 %1, %1, %4, %5\n s_or_b64 exec, exec, %3\n"
                              : "+v"(a0), "+v"(a1), "+s"(m0), "+s"(m1) : "v"(x), "v"(y) : "vcc", "scc");)
        }
    }
    const unsigned long long t1 = clock64();
    const unsigned long long r1 = wall_clock64();
    if (MODE == 12) a0 += lds[tid & 1023].x;
    out[tid] = (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7) ^ u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7 ^ s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7
               ^ (uint32_t)(d0.x + d1.y + d2.x + d3.y) ^ (uint32_t)(m0 ^ m1 ^ m2 ^ m3) ^ (uint32_t)(w0 ^ w1 ^ w2 ^ w3);
    if (tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

static int WPS = 4, TPB = 1024;
template <int MODE>
int run(const char* name, int vper, int sper, uint32_t* out, unsigned long long* clk)
{
    const int blocks = 256, iters = 2000;   // one block of 16 waves per CU = 4 waves per SIMD (WPS=4)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    bench<MODE><<<blocks, TPB>>>(out, 10, 1u, clk);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    bench<MODE><<<blocks, TPB>>>(out, iters, 1u, clk);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double groups = 8.0 * iters;
    // wave 0's own cycle count / instructions it issued, times... 4 waves share the SIMD
    const double cyc = (double)h[0];
    printf("[%5.2f clk/instr/SIMD by kernel time at 2.4 GHz] ", ms * 1e-3 * 2.4e9 / (groups * (vper + sper) * (TPB / 256.0)));
    printf("%-44s %7.3f ms  wave0: %8.0f clk (%.0f MHz)  per group: %6.1f clk -> %5.2f clk per instr per SIMD (VALU %d SALU %d / group)\n",
           name, ms, cyc, cyc / ((double)h[1] / 100.0), cyc / groups, cyc / groups / (vper + sper) / (TPB / 256.0), vper, sper);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc > 1) { TPB = atoi(argv[1]); WPS = 4; }    // waves per SIMD = TPB / 256 with one block per CU
    printf("threads per block %d (one block per CU): %d waves per SIMD\n", TPB, TPB / 256);
    uint32_t* out; unsigned long long* clk;
    CK(hipMalloc(&out, 256 * 4 * 256 * 4)); CK(hipMalloc(&clk, 16));
    run<0>("v_fma_f32", 8, 0, out, clk);
    run<10>("v_mul_f32 (VOP2)", 8, 0, out, clk);
    run<1>("v_xor_b32", 8, 0, out, clk);
    run<2>("v_mad_u64_u32", 8, 0, out, clk);
    run<6>("v_cmp_lt_f32 -> sgpr", 8, 0, out, clk);
    run<7>("v_cndmask_b32 (sgpr mask)", 8, 0, out, clk);
    run<8>("v_cvt_f32_u32", 8, 0, out, clk);
    run<9>("v_mbcnt_lo/hi", 8, 0, out, clk);
    run<3>("s_add_u32", 0, 8, out, clk);
    run<4>("8 v_fma + 8 s_add interleaved", 8, 8, out, clk);
    run<11>("8 v_fma + 16 s_add interleaved", 8, 16, out, clk);
    run<5>("8 v_fma + 4 s_and_b64", 8, 4, out, clk);
    run<13>("2x(v_cmp, saveexec, v_fma, s_or exec)", 4, 4, out, clk);
    run<12>("ds_write_b128 x8 + wait", 8, 0, out, clk);
    run<32>("ds_write_b64 x8 + wait", 8, 0, out, clk);
    run<33>("8 v_fma + 4 ds_write_b128", 8, 4, out, clk);
    run<20>("v_fmac_f32 (VOP2)", 8, 0, out, clk);
    run<21>("v_fmaak_f32 (VOP2+literal)", 8, 0, out, clk);
    run<22>("v_cmp_lt_f32 e32 -> vcc", 8, 0, out, clk);
    run<23>("v_cndmask_b32 e32 (vcc)", 8, 0, out, clk);
    run<24>("v_add_u32 (VOP2)", 8, 0, out, clk);
    run<25>("v_lshl_add_u32 (VOP3)", 8, 0, out, clk);
    run<26>("v_xor_b32 sgpr src", 8, 0, out, clk);
    run<27>("v_mad_u64_u32 sgpr multiplier", 8, 0, out, clk);
    run<28>("v_mul_hi_u32 / v_mul_lo_u32", 8, 0, out, clk);
    run<29>("v_pk_mul_f32", 8, 0, out, clk);
    run<30>("v_addc_co_u32 f+f+bit", 8, 0, out, clk);
    run<31>("v_max3_f32", 8, 0, out, clk);
    return 0;
}
