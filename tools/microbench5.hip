// Issue cost per instruction kind on gfx950, measured so that the two columns of microbench4
// can be reconciled: every wave stamps its own loop (s_memtime + s_memrealtime), and the table
// shows the fastest wave next to the slowest, the median, and the whole kernel.
//   hipcc --offload-arch=gfx950 -O3 -o microbench5 microbench5.hip ; ./microbench5 [threads per block]
// One block per CU (256 blocks); threads per block / 256 = waves per SIMD.
// Register environment of every body: %0-%7 eight independent VGPRs (read-write), two VGPR
// inputs (%13 %14), two SGPRs (%8 %9), two SGPR pairs (%10 %11), a 16-byte VGPR tuple (%12), an LDS byte address (%15).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define REP8(X) X X X X X X X X
#define I8(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
// two kinds alternating: A on even registers, B on odd ones
#define AB8(A, B) A(0) B(1) A(2) B(3) A(4) B(5) A(6) B(7)

struct Stamp { unsigned long long cyc, t0, t1; };   // loop cycles (s_memtime), start/end in 100 MHz ticks

#define KERNEL(NAME, BODY)                                                                                 \
    __global__ __launch_bounds__(1024) void k_##NAME(uint32_t* out, int iters, uint32_t seed, Stamp* st)  \
    {                                                                                                      \
        const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;                                        \
        uint32_t r0 = 0x3f800100u + tid, r1 = 0x3f810000u + tid, r2 = 0x3f820000u + tid, r3 = 0x3f830000u + tid, \
                 r4 = 0x3f840000u + tid, r5 = 0x3f850000u + tid, r6 = 0x3f860000u + tid, r7 = 0x3f870000u + tid; \
        uint32_t x = 0x3f800001u, y = 0x33000000u + (tid & 1u);                                             \
        uint32_t sa = seed, sb = seed * 3u;                                                                \
        unsigned long long m0 = seed * 0x9E3779B97F4A7C15ull, m1 = ~m0;                                    \
        __shared__ uint4 lds[2048];                                                                        \
        lds[threadIdx.x] = make_uint4(tid, 0x3f800000u, 0x3f000000u, 0x3eaaaaabu);                         \
        lds[threadIdx.x + 1024] = make_uint4(tid, 0x3f800000u, 0x3f000000u, 0x3eaaaaabu);                  \
        __syncthreads();                                                                                   \
        const uint32_t la = (threadIdx.x & 63u) * 16u + (threadIdx.x >> 6) * 1024u;                        \
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));                                        \
        u32x4 q4 = {tid, x, y, sa};                                                                        \
        const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime(); \
        for (int i = 0; i < iters; ++i) {                                                                  \
            REP8(asm volatile(BODY : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), \
                                "+s"(sa), "+s"(sb), "+s"(m0), "+s"(m1), "+v"(q4)                                       \
                              : "v"(x), "v"(y), "v"(la) : "vcc", "scc", "memory");) \
        }                                                                                                  \
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime(); \
        out[tid] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ (uint32_t)m0 ^ (uint32_t)m1 ^ lds[(tid * 7) & 2047].x ^ q4.x; \
        if ((threadIdx.x & 63) == 0) {                                                                     \
            Stamp s; s.cyc = c1 - c0; s.t0 = w0; s.t1 = w1;                                                \
            st[tid >> 6] = s;                                                                              \
        }                                                                                                  \
    }

#define S(x) #x
// ---- instruction forms (n = register number) -------------------------------------------------------
#define F_MUL(n)      "v_mul_f32 %" S(n) ", %" S(n) ", %13\n"
#define F_ADD(n)      "v_add_f32 %" S(n) ", %" S(n) ", %13\n"
#define F_FMAC(n)     "v_fmac_f32 %" S(n) ", %13, %14\n"
#define F_FMAAK(n)    "v_fmaak_f32 %" S(n) ", %" S(n) ", %13, 0x3f800001\n"
#define F_FMAMK(n)    "v_fmamk_f32 %" S(n) ", %" S(n) ", 0x3f800001, %13\n"
#define F_FMA(n)      "v_fma_f32 %" S(n) ", %" S(n) ", %13, %14\n"
#define F_FMANEG(n)   "v_fma_f32 %" S(n) ", -%" S(n) ", %13, 1.0\n"
#define F_MULS(n)     "v_mul_f32 %" S(n) ", %8, %" S(n) "\n"
#define F_MULLIT(n)   "v_mul_f32 %" S(n) ", 0x3f800001, %" S(n) "\n"
#define F_MULINL(n)   "v_mul_f32 %" S(n) ", 0.5, %" S(n) "\n"
#define F_MAXF(n)     "v_max_f32 %" S(n) ", %" S(n) ", %13\n"
#define F_XOR(n)      "v_xor_b32 %" S(n) ", %" S(n) ", %13\n"
#define F_OR(n)       "v_or_b32 %" S(n) ", %" S(n) ", %13\n"
#define F_ORINL(n)    "v_or_b32 %" S(n) ", 3, %" S(n) "\n"
#define F_AND(n)      "v_and_b32 %" S(n) ", %" S(n) ", %13\n"
#define F_ANDLIT(n)   "v_and_b32 %" S(n) ", 0xffffff00, %" S(n) "\n"
#define F_ADDU(n)     "v_add_u32 %" S(n) ", %" S(n) ", %13\n"
#define F_ADDUS(n)    "v_add_u32 %" S(n) ", %8, %" S(n) "\n"
#define F_SUBU(n)     "v_sub_u32 %" S(n) ", %" S(n) ", %13\n"
#define F_LSHR(n)     "v_lshrrev_b32 %" S(n) ", 9, %" S(n) "\n"
#define F_LSHL(n)     "v_lshlrev_b32 %" S(n) ", 4, %" S(n) "\n"
#define F_MINU(n)     "v_min_u32 %" S(n) ", %" S(n) ", %13\n"
#define F_MOV(n)      "v_mov_b32 %" S(n) ", %13\n"
#define F_BITOP3(n)   "v_bitop3_b32 %" S(n) ", %" S(n) ", %13, %14 bitop3:0x96\n"
#define F_BITOP3S(n)  "v_bitop3_b32 %" S(n) ", %" S(n) ", %13, %8 bitop3:0x96\n"
#define F_CVTFU(n)    "v_cvt_f32_u32 %" S(n) ", %" S(n) "\n"
#define F_CVTUF(n)    "v_cvt_u32_f32 %" S(n) ", %" S(n) "\n"
#define F_LOG(n)      "v_log_f32 %" S(n) ", %" S(n) "\n"
#define F_EXP(n)      "v_exp_f32 %" S(n) ", %" S(n) "\n"
#define F_RCP(n)      "v_rcp_f32 %" S(n) ", %" S(n) "\n"
#define F_MIN3U(n)    "v_min3_u32 %" S(n) ", %" S(n) ", %13, %14\n"
#define F_LSHLADD(n)  "v_lshl_add_u32 %" S(n) ", %" S(n) ", 4, %13\n"
#define F_LSHLOR(n)   "v_lshl_or_b32 %" S(n) ", %" S(n) ", 4, %13\n"
#define F_ANDOR(n)    "v_and_or_b32 %" S(n) ", %" S(n) ", %13, %14\n"
#define F_ADD3(n)     "v_add3_u32 %" S(n) ", %" S(n) ", %13, %14\n"
#define F_BFE(n)      "v_bfe_u32 %" S(n) ", %" S(n) ", 8, 8\n"
#define F_ALIGNBIT(n) "v_alignbit_b32 %" S(n) ", %" S(n) ", %13, 9\n"
#define F_PERM(n)     "v_perm_b32 %" S(n) ", %" S(n) ", %13, %14\n"
#define F_MULU24(n)   "v_mul_u32_u24 %" S(n) ", %" S(n) ", %13\n"
#define F_MADU24(n)   "v_mad_u32_u24 %" S(n) ", %" S(n) ", %13, %14\n"
#define F_MULLO(n)    "v_mul_lo_u32 %" S(n) ", %" S(n) ", %13\n"
#define F_MULHI(n)    "v_mul_hi_u32 %" S(n) ", %" S(n) ", %13\n"
#define F_CMPS(n)     "v_cmp_lt_f32 %10, %" S(n) ", %13\n"
#define F_CMPUS(n)    "v_cmp_lt_u32 %10, %" S(n) ", %13\n"
#define F_CMPVCC(n)   "v_cmp_lt_f32 vcc, %" S(n) ", %13\n"
#define F_CNDS(n)     "v_cndmask_b32 %" S(n) ", %" S(n) ", %13, %10\n"
#define F_MBCNTLO(n)  "v_mbcnt_lo_u32_b32 %" S(n) ", %8, %" S(n) "\n"
#define F_MBCNTHI(n)  "v_mbcnt_hi_u32_b32 %" S(n) ", %9, %" S(n) "\n"
#define F_SADD(n)     "s_add_u32 %8, %8, %9\n"
#define F_SAND64(n)   "s_and_b64 %10, %10, %11\n"
#define F_SBCNT(n)    "s_bcnt1_i32_b64 %8, %10\n"
#define F_SMOVEXEC(n) "s_mov_b64 exec, -1\n"
#define F_DSR128(n)   "ds_read_b128 %12, %15\n"
#define F_DSW128(n)   "ds_write_b128 %15, %12\n"
#define F_DSW128M(n)  "s_mov_b64 exec, %10\nds_write_b128 %15, %12\ns_mov_b64 exec, -1\n"
#define F_DSW64(n)    "ds_write_b64 %15, %12\n"
#define F_DSW32(n)    "ds_write_b32 %15, %13\n"
#define F_DSW8(n)     "ds_write_b8 %15, %13\n"
#define F_DSR32(n)    "ds_read_b32 %" S(n) ", %15\n"
#define F_NOP(n)      "s_nop 0\n"

// the stage-1 push of the stream kernel, one sample: cmp -> sgpr, 2 mbcnt, lshl_add, masked 16-B store, bcnt, add
#define F_PUSH(n)     "v_cmp_lt_f32 %10, %" S(n) ", %13\nv_mbcnt_lo_u32_b32 %" S(n) ", %8, 0\nv_mbcnt_hi_u32_b32 %" S(n) ", %9, %" S(n) "\n" \
                      "v_lshl_add_u32 %" S(n) ", %" S(n) ", 4, %13\n" \
                      "s_mov_b64 exec, %11\nds_write_b128 %15, %12\ns_mov_b64 exec, -1\ns_bcnt1_i32_b64 %8, %10\ns_add_u32 %9, %9, %8\n"

#define LIST(X) \
    X(mul_f32, I8(F_MUL), 8, "v_mul_f32 (VOP2, VGPRs)") \
    X(add_f32, I8(F_ADD), 8, "v_add_f32") \
    X(fmac_f32, I8(F_FMAC), 8, "v_fmac_f32 (VOP2)") \
    X(fmaak_f32, I8(F_FMAAK), 8, "v_fmaak_f32 (VOP2 + literal)") \
    X(fmamk_f32, I8(F_FMAMK), 8, "v_fmamk_f32 (VOP2 + literal)") \
    X(fma_f32, I8(F_FMA), 8, "v_fma_f32 (VOP3)") \
    X(fma_neg, I8(F_FMANEG), 8, "v_fma_f32 -a, b, 1.0 (VOP3 + modifier)") \
    X(mul_sgpr, I8(F_MULS), 8, "v_mul_f32 with an SGPR source") \
    X(mul_lit, I8(F_MULLIT), 8, "v_mul_f32 with a literal") \
    X(mul_inl, I8(F_MULINL), 8, "v_mul_f32 with an inline constant") \
    X(max_f32, I8(F_MAXF), 8, "v_max_f32") \
    X(xor_b32, I8(F_XOR), 8, "v_xor_b32") \
    X(or_b32, I8(F_OR), 8, "v_or_b32") \
    X(or_inl, I8(F_ORINL), 8, "v_or_b32 with an inline constant") \
    X(and_b32, I8(F_AND), 8, "v_and_b32") \
    X(and_lit, I8(F_ANDLIT), 8, "v_and_b32 with a literal") \
    X(add_u32, I8(F_ADDU), 8, "v_add_u32") \
    X(add_u32_s, I8(F_ADDUS), 8, "v_add_u32 with an SGPR source") \
    X(sub_u32, I8(F_SUBU), 8, "v_sub_u32") \
    X(lshrrev, I8(F_LSHR), 8, "v_lshrrev_b32 (inline shift)") \
    X(lshlrev, I8(F_LSHL), 8, "v_lshlrev_b32 (inline shift)") \
    X(min_u32, I8(F_MINU), 8, "v_min_u32") \
    X(mov_b32, I8(F_MOV), 8, "v_mov_b32") \
    X(bitop3, I8(F_BITOP3), 8, "v_bitop3_b32 (VGPRs)") \
    X(bitop3_s, I8(F_BITOP3S), 8, "v_bitop3_b32 with an SGPR source") \
    X(cvt_f32_u32, I8(F_CVTFU), 8, "v_cvt_f32_u32") \
    X(cvt_u32_f32, I8(F_CVTUF), 8, "v_cvt_u32_f32") \
    X(log_f32, I8(F_LOG), 8, "v_log_f32") \
    X(exp_f32, I8(F_EXP), 8, "v_exp_f32") \
    X(rcp_f32, I8(F_RCP), 8, "v_rcp_f32") \
    X(min3_u32, I8(F_MIN3U), 8, "v_min3_u32") \
    X(lshl_add, I8(F_LSHLADD), 8, "v_lshl_add_u32") \
    X(lshl_or, I8(F_LSHLOR), 8, "v_lshl_or_b32") \
    X(and_or, I8(F_ANDOR), 8, "v_and_or_b32") \
    X(add3, I8(F_ADD3), 8, "v_add3_u32") \
    X(bfe, I8(F_BFE), 8, "v_bfe_u32") \
    X(alignbit, I8(F_ALIGNBIT), 8, "v_alignbit_b32") \
    X(perm, I8(F_PERM), 8, "v_perm_b32") \
    X(mul_u24, I8(F_MULU24), 8, "v_mul_u32_u24 (VOP2)") \
    X(mad_u24, I8(F_MADU24), 8, "v_mad_u32_u24") \
    X(mul_lo, I8(F_MULLO), 8, "v_mul_lo_u32") \
    X(mul_hi, I8(F_MULHI), 8, "v_mul_hi_u32") \
    X(cmp_s, I8(F_CMPS), 8, "v_cmp_lt_f32 -> SGPR pair") \
    X(cmp_u_s, I8(F_CMPUS), 8, "v_cmp_lt_u32 -> SGPR pair") \
    X(cmp_vcc, I8(F_CMPVCC), 8, "v_cmp_lt_f32 -> vcc (e32)") \
    X(cnd_s, I8(F_CNDS), 8, "v_cndmask_b32, SGPR-pair mask") \
    X(mbcnt, AB8(F_MBCNTLO, F_MBCNTHI), 8, "v_mbcnt_lo / v_mbcnt_hi") \
    X(s_add, I8(F_SADD), 8, "s_add_u32") \
    X(s_and64, I8(F_SAND64), 8, "s_and_b64") \
    X(ds_read_b128, I8(F_DSR128) "s_waitcnt lgkmcnt(0)\n", 8, "ds_read_b128 x8 + wait") \
    X(ds_write_b128, I8(F_DSW128) "s_waitcnt lgkmcnt(0)\n", 8, "ds_write_b128 x8 + wait") \
    X(ds_write_b128m, I8(F_DSW128M) "s_waitcnt lgkmcnt(0)\n", 8, "ds_write_b128 under a ~50 % exec mask x8 + wait") \
    X(ds_write_b32, I8(F_DSW32) "s_waitcnt lgkmcnt(0)\n", 8, "ds_write_b32 x8 + wait") \
    X(ds_write_b8, I8(F_DSW8) "s_waitcnt lgkmcnt(0)\n", 8, "ds_write_b8 x8 + wait") \
    X(mix_mul_cvt, AB8(F_MUL, F_CVTFU), 8, "4 v_mul_f32 + 4 v_cvt_f32_u32 alternating") \
    X(mix_mul_mad, AB8(F_MUL, F_MULHI), 8, "4 v_mul_f32 + 4 v_mul_hi_u32 alternating") \
    X(mix_fma_sadd, AB8(F_FMAC, F_SADD), 8, "4 v_fmac + 4 s_add alternating") \
    X(mix_fmac2_sadd, F_FMAC(0) F_FMAC(1) F_SADD(0) F_FMAC(2) F_FMAC(3) F_SADD(0) F_FMAC(4) F_FMAC(5) F_SADD(0) F_FMAC(6) F_FMAC(7) F_SADD(0), 12, "8 v_fmac + 4 s_add") \
    X(mix_cvt_sadd, AB8(F_CVTFU, F_SADD), 8, "4 v_cvt + 4 s_add alternating") \
    X(mix_mul_dsw, F_MUL(0) F_MUL(1) F_MUL(2) F_DSW128(0) F_MUL(3) F_MUL(4) F_MUL(5) F_DSW128(0) "s_waitcnt lgkmcnt(0)\n", 8, "6 v_mul + 2 ds_write_b128 + wait") \
    X(push, F_PUSH(0) F_PUSH(1) F_PUSH(2) F_PUSH(3) "s_waitcnt lgkmcnt(0)\n", 36, "stage-1 push x4 (cmp, 2 mbcnt, lshl_add, masked ds_write_b128, bcnt, add)") \
    X(nop, I8(F_NOP), 8, "s_nop 0")

#define DEF(NAME, BODY, N, DESC) KERNEL(NAME, BODY)
LIST(DEF)

typedef void (*kern_t)(uint32_t*, int, uint32_t, Stamp*);
struct Entry { const char* name; kern_t fn; int per_group; const char* desc; };
#define ENT(NAME, BODY, N, DESC) {#NAME, k_##NAME, N, DESC},
static Entry entries[] = { LIST(ENT) };

int main(int argc, char** argv)
{
    const int tpb = argc > 1 ? atoi(argv[1]) : 1024;
    const char* only = (argc > 2 && argv[2][0]) ? argv[2] : nullptr;
    const int per_cu = argc > 3 ? atoi(argv[3]) : 1;          // blocks per CU (2 x 1024 threads: 8 waves per SIMD)
    const int blocks = 256 * per_cu, iters = 2000, waves = blocks * tpb / 64;
    printf("# %d threads per block, %d block(s) per CU: %d waves per SIMD; %d groups of 8 statements per wave\n", tpb, per_cu, per_cu * tpb / 256, iters);
    printf("# cycles per instruction per SIMD = wave loop cycles / instructions per wave / waves per SIMD\n");
    printf("# %-58s %8s | %7s %7s %7s | %7s | %6s %6s\n", "kind", "kern ms", "fastest", "median", "slowest", "kernel", "MHz", "spread");
    uint32_t* out; Stamp* st;
    CK(hipMalloc(&out, (size_t)blocks * 1024 * 4)); CK(hipMalloc(&st, sizeof(Stamp) * waves));
    std::vector<Stamp> h(waves);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Entry& e : entries) {
        if (only && strcmp(only, e.name) != 0) continue;
        e.fn<<<blocks, tpb>>>(out, 10, 1u, st);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        e.fn<<<blocks, tpb>>>(out, iters, 1u, st);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), st, sizeof(Stamp) * waves, hipMemcpyDeviceToHost));
        std::vector<double> cyc(waves);
        unsigned long long tmin = ~0ull, tmax = 0; double mhz = 0;
        for (int i = 0; i < waves; ++i) {
            cyc[i] = (double)h[i].cyc;
            tmin = std::min(tmin, h[i].t0); tmax = std::max(tmax, h[i].t1);
            mhz += (double)h[i].cyc / ((double)(h[i].t1 - h[i].t0) / 100.0);
        }
        mhz /= waves;
        std::sort(cyc.begin(), cyc.end());
        const double instr = 8.0 * iters * e.per_group, wps = per_cu * tpb / 256.0;
        // whole kernel: first loop start to last loop end, in shader cycles at the measured clock
        const double kern_cyc = (double)(tmax - tmin) / 100.0 * mhz;
        printf("  %-58s %8.3f | %7.2f %7.2f %7.2f | %7.2f | %6.0f %6.2f\n", e.desc, ms,
               cyc.front() / instr / wps, cyc[waves / 2] / instr / wps, cyc.back() / instr / wps,
               kern_cyc / instr / wps, mhz, cyc.back() / cyc.front());
    }
    return 0;
}
