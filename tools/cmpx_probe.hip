// What v_cmpx_*_e32 writes on gfx950 (exec? vcc?), and whether the instruction behind it sees the new exec.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(unsigned long long* out, uint32_t* ranks)
{
    const int lane = threadIdx.x;
    const float x = lane % 3 == 0 ? 1.0f : 3.0f, two = 2.0f;
    unsigned long long ex, vc;
    uint32_t rank = 0xffffffffu, n;
    asm volatile("s_mov_b64 vcc, 0\n\t"
                 "v_cmpx_lt_f32_e32 vcc, %[x], %[two]\n\t"
                 "v_mbcnt_lo_u32_b32 %[r], exec_lo, 0\n\t"
                 "v_mbcnt_hi_u32_b32 %[r], exec_hi, %[r]\n\t"
                 "s_mov_b64 %[ex], exec\n\t"
                 "s_mov_b64 %[vc], vcc\n\t"
                 "s_bcnt1_i32_b64 %[n], exec\n\t"
                 "s_mov_b64 exec, -1"
                 : [r] "+v"(rank), [ex] "=&s"(ex), [vc] "=&s"(vc), [n] "=&s"(n) : [x] "v"(x), [two] "v"(two) : "vcc", "scc");
    ranks[lane] = rank;
    if (lane == 0) { out[0] = ex; out[1] = vc; out[2] = n; }
}
int main()
{
    unsigned long long* d; uint32_t* r;
    hipMalloc(&d, 24); hipMalloc(&r, 256);
    probe<<<1, 64>>>(d, r);
    unsigned long long h[3]; uint32_t hr[64];
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost); hipMemcpy(hr, r, 256, hipMemcpyDeviceToHost);
    printf("exec after v_cmpx %016llx  vcc %016llx  bcnt %llu\n", h[0], h[1], h[2]);
    for (int i = 0; i < 16; ++i) printf("%d:%x ", i, hr[i]);
    printf("\n");
    return 0;
}
