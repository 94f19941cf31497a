// Which VALU instruction kinds share an issue slot on gfx950: every kind alone, alternating with v_mul_f32
// (a full-rate VOP2), alternating with v_cvt_f32_u32 (a 'slow' kind), and as a dependent chain on one
// register.  Same harness as microbench5 (generated from it by the script in its header comment).
//   hipcc --offload-arch=gfx950 -O3 -o microbench6 microbench6.hip ; ./microbench6 [threads per block]
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
        unsigned long long ww0, ww1, ww2, ww3; asm volatile("v_mov_b64 %0, 1\n v_mov_b64 %1, 2\n v_mov_b64 %2, 3\n v_mov_b64 %3, 4" : "=v"(ww0), "=v"(ww1), "=v"(ww2), "=v"(ww3));                           \
        const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime(); \
        for (int i = 0; i < iters; ++i) {                                                                  \
            REP8(asm volatile(BODY : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), \
                                "+s"(sa), "+s"(sb), "+s"(m0), "+s"(m1), "+v"(q4)                                       \
                              : "v"(x), "v"(y), "v"(la), "v"(ww0), "v"(ww1), "v"(ww2), "v"(ww3) : "vcc", "scc", "memory");) \
        }                                                                                                  \
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime(); \
        out[tid] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ (uint32_t)m0 ^ (uint32_t)m1 ^ lds[(tid * 7) & 2047].x ^ q4.x ^ (uint32_t)(ww0 ^ ww1 ^ ww2 ^ ww3); \
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

#define F_MAD64_0 "v_mad_u64_u32 %16, vcc, %0, %13, 0\n"
#define F_MAD64_1 "v_mad_u64_u32 %17, vcc, %1, %13, 0\n"
#define F_MAD64_2 "v_mad_u64_u32 %18, vcc, %2, %13, 0\n"
#define F_MAD64_3 "v_mad_u64_u32 %19, vcc, %3, %13, 0\n"
#define F_MAD64_4 "v_mad_u64_u32 %16, vcc, %4, %13, 0\n"
#define F_MAD64_5 "v_mad_u64_u32 %17, vcc, %5, %13, 0\n"
#define F_MAD64_6 "v_mad_u64_u32 %18, vcc, %6, %13, 0\n"
#define F_MAD64_7 "v_mad_u64_u32 %19, vcc, %7, %13, 0\n"
#define F_MAD64(n) F_MAD64_##n
#define F_SUBF(n)     "v_sub_f32 %" S(n) ", %" S(n) ", %13\n"
#define F_MINF(n)     "v_min_f32 %" S(n) ", %" S(n) ", %13\n"
#define F_MAXU(n)     "v_max_u32 %" S(n) ", %" S(n) ", %13\n"
#define F_ASHR(n)     "v_ashrrev_i32 %" S(n) ", 31, %" S(n) "\n"
#define F_LSHLV(n)    "v_lshlrev_b32 %" S(n) ", %13, %" S(n) "\n"
#define F_SUBREV(n)   "v_subrev_u32 %" S(n) ", %13, %" S(n) "\n"
#define F_CNDVCC(n)   "v_cndmask_b32 %" S(n) ", %" S(n) ", %13, vcc\n"
#define F_ADDC(n)     "v_addc_co_u32 %" S(n) ", vcc, %" S(n) ", %13, vcc\n"
#define F_FMA2(n)     "v_fma_f32 %" S(n) ", %" S(n) ", %13, %13\n"
#define F_MED3(n)     "v_med3_f32 %" S(n) ", %" S(n) ", %13, %14\n"
#define F_SAD(n)      "v_sad_u32 %" S(n) ", %" S(n) ", %13, %14\n"
#define F_XAD(n)      "v_xad_u32 %" S(n) ", %" S(n) ", %13, %14\n"
#define F_MULLEG(n)   "v_mul_legacy_f32 %" S(n) ", %" S(n) ", %13\n"
#define F_CVTI(n)     "v_cvt_f32_i32 %" S(n) ", %" S(n) "\n"
#define F_FLOOR(n)    "v_floor_f32 %" S(n) ", %" S(n) "\n"
#define F_FRACT(n)    "v_fract_f32 %" S(n) ", %" S(n) "\n"
#define F_LDEXP(n)    "v_ldexp_f32 %" S(n) ", %" S(n) ", %13\n"
#define F_READLANE(n) "v_readlane_b32 %8, %" S(n) ", 5\n"
#define F_DPP(n)      "v_mov_b32_dpp %" S(n) ", %" S(n) " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define F_SDWA(n)     "v_and_b32_sdwa %" S(n) ", %" S(n) ", %13 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define F_MULCL(n)    "v_mul_f32_e64 %" S(n) ", %" S(n) ", %13 clamp\n"
#define F_FMAABS(n)   "v_fma_f32 %" S(n) ", |%" S(n) "|, %13, 1.0 clamp\n"
#define F_ADDABS(n)   "v_add_f32_e64 %" S(n) ", |%" S(n) "|, %13\n"
#define F_CMPABS(n)   "v_cmp_lt_f32_e64 %10, |%" S(n) "|, %13\n"
#define F_CMPXF(n)    "v_cmp_gt_f32 vcc, 0, %" S(n) "\n"
#define F_MIN3F(n)    "v_min3_f32 %" S(n) ", %" S(n) ", %13, %14\n"
#define F_ANDOR2(n)   "v_and_or_b32 %" S(n) ", %" S(n) ", %13, %14\n"
// dependent chains: all eight statements of a group on register 0
#define D8(F) F(0) F(0) F(0) F(0) F(0) F(0) F(0) F(0)

#define LIST(X) \
    X(a_mul, I8(F_MUL), 8, "v_mul_f32 alone") \
    X(c_mul, AB8(F_CVTFU, F_MUL), 8, "v_mul_f32 | v_cvt_f32_u32") \
    X(d_mul, D8(F_MUL), 8, "v_mul_f32 dependent chain") \
    X(a_add, I8(F_ADD), 8, "v_add_f32 alone") \
    X(m_add, AB8(F_MUL, F_ADD), 8, "v_add_f32 | v_mul_f32") \
    X(c_add, AB8(F_CVTFU, F_ADD), 8, "v_add_f32 | v_cvt_f32_u32") \
    X(d_add, D8(F_ADD), 8, "v_add_f32 dependent chain") \
    X(a_subf, I8(F_SUBF), 8, "v_sub_f32 alone") \
    X(m_subf, AB8(F_MUL, F_SUBF), 8, "v_sub_f32 | v_mul_f32") \
    X(c_subf, AB8(F_CVTFU, F_SUBF), 8, "v_sub_f32 | v_cvt_f32_u32") \
    X(d_subf, D8(F_SUBF), 8, "v_sub_f32 dependent chain") \
    X(a_fmaak, I8(F_FMAAK), 8, "v_fmaak_f32 alone") \
    X(m_fmaak, AB8(F_MUL, F_FMAAK), 8, "v_fmaak_f32 | v_mul_f32") \
    X(c_fmaak, AB8(F_CVTFU, F_FMAAK), 8, "v_fmaak_f32 | v_cvt_f32_u32") \
    X(d_fmaak, D8(F_FMAAK), 8, "v_fmaak_f32 dependent chain") \
    X(a_fmac, I8(F_FMAC), 8, "v_fmac_f32 (3 VGPRs read) alone") \
    X(m_fmac, AB8(F_MUL, F_FMAC), 8, "v_fmac_f32 (3 VGPRs read) | v_mul_f32") \
    X(c_fmac, AB8(F_CVTFU, F_FMAC), 8, "v_fmac_f32 (3 VGPRs read) | v_cvt_f32_u32") \
    X(d_fmac, D8(F_FMAC), 8, "v_fmac_f32 (3 VGPRs read) dependent chain") \
    X(a_fma3, I8(F_FMA), 8, "v_fma_f32 a,b,c (3 VGPRs) alone") \
    X(m_fma3, AB8(F_MUL, F_FMA), 8, "v_fma_f32 a,b,c (3 VGPRs) | v_mul_f32") \
    X(c_fma3, AB8(F_CVTFU, F_FMA), 8, "v_fma_f32 a,b,c (3 VGPRs) | v_cvt_f32_u32") \
    X(d_fma3, D8(F_FMA), 8, "v_fma_f32 a,b,c (3 VGPRs) dependent chain") \
    X(a_fma2, I8(F_FMA2), 8, "v_fma_f32 a,b,b (2 VGPRs) alone") \
    X(m_fma2, AB8(F_MUL, F_FMA2), 8, "v_fma_f32 a,b,b (2 VGPRs) | v_mul_f32") \
    X(c_fma2, AB8(F_CVTFU, F_FMA2), 8, "v_fma_f32 a,b,b (2 VGPRs) | v_cvt_f32_u32") \
    X(d_fma2, D8(F_FMA2), 8, "v_fma_f32 a,b,b (2 VGPRs) dependent chain") \
    X(a_maxf, I8(F_MAXF), 8, "v_max_f32 alone") \
    X(m_maxf, AB8(F_MUL, F_MAXF), 8, "v_max_f32 | v_mul_f32") \
    X(c_maxf, AB8(F_CVTFU, F_MAXF), 8, "v_max_f32 | v_cvt_f32_u32") \
    X(d_maxf, D8(F_MAXF), 8, "v_max_f32 dependent chain") \
    X(a_minf, I8(F_MINF), 8, "v_min_f32 alone") \
    X(m_minf, AB8(F_MUL, F_MINF), 8, "v_min_f32 | v_mul_f32") \
    X(c_minf, AB8(F_CVTFU, F_MINF), 8, "v_min_f32 | v_cvt_f32_u32") \
    X(d_minf, D8(F_MINF), 8, "v_min_f32 dependent chain") \
    X(a_med3, I8(F_MED3), 8, "v_med3_f32 alone") \
    X(m_med3, AB8(F_MUL, F_MED3), 8, "v_med3_f32 | v_mul_f32") \
    X(c_med3, AB8(F_CVTFU, F_MED3), 8, "v_med3_f32 | v_cvt_f32_u32") \
    X(d_med3, D8(F_MED3), 8, "v_med3_f32 dependent chain") \
    X(a_xor, I8(F_XOR), 8, "v_xor_b32 alone") \
    X(m_xor, AB8(F_MUL, F_XOR), 8, "v_xor_b32 | v_mul_f32") \
    X(c_xor, AB8(F_CVTFU, F_XOR), 8, "v_xor_b32 | v_cvt_f32_u32") \
    X(d_xor, D8(F_XOR), 8, "v_xor_b32 dependent chain") \
    X(a_addu, I8(F_ADDU), 8, "v_add_u32 alone") \
    X(m_addu, AB8(F_MUL, F_ADDU), 8, "v_add_u32 | v_mul_f32") \
    X(c_addu, AB8(F_CVTFU, F_ADDU), 8, "v_add_u32 | v_cvt_f32_u32") \
    X(d_addu, D8(F_ADDU), 8, "v_add_u32 dependent chain") \
    X(a_subu, I8(F_SUBU), 8, "v_sub_u32 alone") \
    X(m_subu, AB8(F_MUL, F_SUBU), 8, "v_sub_u32 | v_mul_f32") \
    X(c_subu, AB8(F_CVTFU, F_SUBU), 8, "v_sub_u32 | v_cvt_f32_u32") \
    X(d_subu, D8(F_SUBU), 8, "v_sub_u32 dependent chain") \
    X(a_subrev, I8(F_SUBREV), 8, "v_subrev_u32 alone") \
    X(m_subrev, AB8(F_MUL, F_SUBREV), 8, "v_subrev_u32 | v_mul_f32") \
    X(c_subrev, AB8(F_CVTFU, F_SUBREV), 8, "v_subrev_u32 | v_cvt_f32_u32") \
    X(d_subrev, D8(F_SUBREV), 8, "v_subrev_u32 dependent chain") \
    X(a_lshr, I8(F_LSHR), 8, "v_lshrrev_b32 const alone") \
    X(m_lshr, AB8(F_MUL, F_LSHR), 8, "v_lshrrev_b32 const | v_mul_f32") \
    X(c_lshr, AB8(F_CVTFU, F_LSHR), 8, "v_lshrrev_b32 const | v_cvt_f32_u32") \
    X(d_lshr, D8(F_LSHR), 8, "v_lshrrev_b32 const dependent chain") \
    X(a_lshl, I8(F_LSHL), 8, "v_lshlrev_b32 const alone") \
    X(m_lshl, AB8(F_MUL, F_LSHL), 8, "v_lshlrev_b32 const | v_mul_f32") \
    X(c_lshl, AB8(F_CVTFU, F_LSHL), 8, "v_lshlrev_b32 const | v_cvt_f32_u32") \
    X(d_lshl, D8(F_LSHL), 8, "v_lshlrev_b32 const dependent chain") \
    X(a_ashr, I8(F_ASHR), 8, "v_ashrrev_i32 31 alone") \
    X(m_ashr, AB8(F_MUL, F_ASHR), 8, "v_ashrrev_i32 31 | v_mul_f32") \
    X(c_ashr, AB8(F_CVTFU, F_ASHR), 8, "v_ashrrev_i32 31 | v_cvt_f32_u32") \
    X(d_ashr, D8(F_ASHR), 8, "v_ashrrev_i32 31 dependent chain") \
    X(a_minu, I8(F_MINU), 8, "v_min_u32 alone") \
    X(m_minu, AB8(F_MUL, F_MINU), 8, "v_min_u32 | v_mul_f32") \
    X(c_minu, AB8(F_CVTFU, F_MINU), 8, "v_min_u32 | v_cvt_f32_u32") \
    X(d_minu, D8(F_MINU), 8, "v_min_u32 dependent chain") \
    X(a_maxu, I8(F_MAXU), 8, "v_max_u32 alone") \
    X(m_maxu, AB8(F_MUL, F_MAXU), 8, "v_max_u32 | v_mul_f32") \
    X(c_maxu, AB8(F_CVTFU, F_MAXU), 8, "v_max_u32 | v_cvt_f32_u32") \
    X(d_maxu, D8(F_MAXU), 8, "v_max_u32 dependent chain") \
    X(a_mov, I8(F_MOV), 8, "v_mov_b32 alone") \
    X(m_mov, AB8(F_MUL, F_MOV), 8, "v_mov_b32 | v_mul_f32") \
    X(c_mov, AB8(F_CVTFU, F_MOV), 8, "v_mov_b32 | v_cvt_f32_u32") \
    X(d_mov, D8(F_MOV), 8, "v_mov_b32 dependent chain") \
    X(a_bitop3, I8(F_BITOP3), 8, "v_bitop3_b32 (3 VGPRs) alone") \
    X(m_bitop3, AB8(F_MUL, F_BITOP3), 8, "v_bitop3_b32 (3 VGPRs) | v_mul_f32") \
    X(c_bitop3, AB8(F_CVTFU, F_BITOP3), 8, "v_bitop3_b32 (3 VGPRs) | v_cvt_f32_u32") \
    X(d_bitop3, D8(F_BITOP3), 8, "v_bitop3_b32 (3 VGPRs) dependent chain") \
    X(a_bitop3s, I8(F_BITOP3S), 8, "v_bitop3_b32 (2 VGPRs + SGPR) alone") \
    X(m_bitop3s, AB8(F_MUL, F_BITOP3S), 8, "v_bitop3_b32 (2 VGPRs + SGPR) | v_mul_f32") \
    X(c_bitop3s, AB8(F_CVTFU, F_BITOP3S), 8, "v_bitop3_b32 (2 VGPRs + SGPR) | v_cvt_f32_u32") \
    X(d_bitop3s, D8(F_BITOP3S), 8, "v_bitop3_b32 (2 VGPRs + SGPR) dependent chain") \
    X(a_mad64, I8(F_MAD64), 8, "v_mad_u64_u32 alone") \
    X(m_mad64, AB8(F_MUL, F_MAD64), 8, "v_mad_u64_u32 | v_mul_f32") \
    X(c_mad64, AB8(F_CVTFU, F_MAD64), 8, "v_mad_u64_u32 | v_cvt_f32_u32") \
    X(a_mulhi, I8(F_MULHI), 8, "v_mul_hi_u32 alone") \
    X(m_mulhi, AB8(F_MUL, F_MULHI), 8, "v_mul_hi_u32 | v_mul_f32") \
    X(c_mulhi, AB8(F_CVTFU, F_MULHI), 8, "v_mul_hi_u32 | v_cvt_f32_u32") \
    X(d_mulhi, D8(F_MULHI), 8, "v_mul_hi_u32 dependent chain") \
    X(a_mullo, I8(F_MULLO), 8, "v_mul_lo_u32 alone") \
    X(m_mullo, AB8(F_MUL, F_MULLO), 8, "v_mul_lo_u32 | v_mul_f32") \
    X(c_mullo, AB8(F_CVTFU, F_MULLO), 8, "v_mul_lo_u32 | v_cvt_f32_u32") \
    X(d_mullo, D8(F_MULLO), 8, "v_mul_lo_u32 dependent chain") \
    X(a_mulu24, I8(F_MULU24), 8, "v_mul_u32_u24 alone") \
    X(m_mulu24, AB8(F_MUL, F_MULU24), 8, "v_mul_u32_u24 | v_mul_f32") \
    X(c_mulu24, AB8(F_CVTFU, F_MULU24), 8, "v_mul_u32_u24 | v_cvt_f32_u32") \
    X(d_mulu24, D8(F_MULU24), 8, "v_mul_u32_u24 dependent chain") \
    X(a_cvtfu, I8(F_CVTFU), 8, "v_cvt_f32_u32 alone") \
    X(m_cvtfu, AB8(F_MUL, F_CVTFU), 8, "v_cvt_f32_u32 | v_mul_f32") \
    X(d_cvtfu, D8(F_CVTFU), 8, "v_cvt_f32_u32 dependent chain") \
    X(a_cvtuf, I8(F_CVTUF), 8, "v_cvt_u32_f32 alone") \
    X(m_cvtuf, AB8(F_MUL, F_CVTUF), 8, "v_cvt_u32_f32 | v_mul_f32") \
    X(c_cvtuf, AB8(F_CVTFU, F_CVTUF), 8, "v_cvt_u32_f32 | v_cvt_f32_u32") \
    X(d_cvtuf, D8(F_CVTUF), 8, "v_cvt_u32_f32 dependent chain") \
    X(a_cvtfi, I8(F_CVTI), 8, "v_cvt_f32_i32 alone") \
    X(m_cvtfi, AB8(F_MUL, F_CVTI), 8, "v_cvt_f32_i32 | v_mul_f32") \
    X(c_cvtfi, AB8(F_CVTFU, F_CVTI), 8, "v_cvt_f32_i32 | v_cvt_f32_u32") \
    X(d_cvtfi, D8(F_CVTI), 8, "v_cvt_f32_i32 dependent chain") \
    X(a_floor, I8(F_FLOOR), 8, "v_floor_f32 alone") \
    X(m_floor, AB8(F_MUL, F_FLOOR), 8, "v_floor_f32 | v_mul_f32") \
    X(c_floor, AB8(F_CVTFU, F_FLOOR), 8, "v_floor_f32 | v_cvt_f32_u32") \
    X(d_floor, D8(F_FLOOR), 8, "v_floor_f32 dependent chain") \
    X(a_ldexp, I8(F_LDEXP), 8, "v_ldexp_f32 alone") \
    X(m_ldexp, AB8(F_MUL, F_LDEXP), 8, "v_ldexp_f32 | v_mul_f32") \
    X(c_ldexp, AB8(F_CVTFU, F_LDEXP), 8, "v_ldexp_f32 | v_cvt_f32_u32") \
    X(d_ldexp, D8(F_LDEXP), 8, "v_ldexp_f32 dependent chain") \
    X(a_log, I8(F_LOG), 8, "v_log_f32 alone") \
    X(m_log, AB8(F_MUL, F_LOG), 8, "v_log_f32 | v_mul_f32") \
    X(c_log, AB8(F_CVTFU, F_LOG), 8, "v_log_f32 | v_cvt_f32_u32") \
    X(d_log, D8(F_LOG), 8, "v_log_f32 dependent chain") \
    X(a_exp, I8(F_EXP), 8, "v_exp_f32 alone") \
    X(m_exp, AB8(F_MUL, F_EXP), 8, "v_exp_f32 | v_mul_f32") \
    X(c_exp, AB8(F_CVTFU, F_EXP), 8, "v_exp_f32 | v_cvt_f32_u32") \
    X(d_exp, D8(F_EXP), 8, "v_exp_f32 dependent chain") \
    X(a_rcp, I8(F_RCP), 8, "v_rcp_f32 alone") \
    X(m_rcp, AB8(F_MUL, F_RCP), 8, "v_rcp_f32 | v_mul_f32") \
    X(c_rcp, AB8(F_CVTFU, F_RCP), 8, "v_rcp_f32 | v_cvt_f32_u32") \
    X(d_rcp, D8(F_RCP), 8, "v_rcp_f32 dependent chain") \
    X(a_cmps, I8(F_CMPS), 8, "v_cmp_lt_f32 -> SGPR alone") \
    X(m_cmps, AB8(F_MUL, F_CMPS), 8, "v_cmp_lt_f32 -> SGPR | v_mul_f32") \
    X(c_cmps, AB8(F_CVTFU, F_CMPS), 8, "v_cmp_lt_f32 -> SGPR | v_cvt_f32_u32") \
    X(d_cmps, D8(F_CMPS), 8, "v_cmp_lt_f32 -> SGPR dependent chain") \
    X(a_cmpvcc, I8(F_CMPVCC), 8, "v_cmp_lt_f32 -> vcc alone") \
    X(m_cmpvcc, AB8(F_MUL, F_CMPVCC), 8, "v_cmp_lt_f32 -> vcc | v_mul_f32") \
    X(c_cmpvcc, AB8(F_CVTFU, F_CMPVCC), 8, "v_cmp_lt_f32 -> vcc | v_cvt_f32_u32") \
    X(d_cmpvcc, D8(F_CMPVCC), 8, "v_cmp_lt_f32 -> vcc dependent chain") \
    X(a_cnds, I8(F_CNDS), 8, "v_cndmask_b32 (SGPR mask) alone") \
    X(m_cnds, AB8(F_MUL, F_CNDS), 8, "v_cndmask_b32 (SGPR mask) | v_mul_f32") \
    X(c_cnds, AB8(F_CVTFU, F_CNDS), 8, "v_cndmask_b32 (SGPR mask) | v_cvt_f32_u32") \
    X(d_cnds, D8(F_CNDS), 8, "v_cndmask_b32 (SGPR mask) dependent chain") \
    X(a_cndvcc, I8(F_CNDVCC), 8, "v_cndmask_b32 (vcc) alone") \
    X(m_cndvcc, AB8(F_MUL, F_CNDVCC), 8, "v_cndmask_b32 (vcc) | v_mul_f32") \
    X(c_cndvcc, AB8(F_CVTFU, F_CNDVCC), 8, "v_cndmask_b32 (vcc) | v_cvt_f32_u32") \
    X(d_cndvcc, D8(F_CNDVCC), 8, "v_cndmask_b32 (vcc) dependent chain") \
    X(a_addc, I8(F_ADDC), 8, "v_addc_co_u32 (vcc in/out) alone") \
    X(m_addc, AB8(F_MUL, F_ADDC), 8, "v_addc_co_u32 (vcc in/out) | v_mul_f32") \
    X(c_addc, AB8(F_CVTFU, F_ADDC), 8, "v_addc_co_u32 (vcc in/out) | v_cvt_f32_u32") \
    X(d_addc, D8(F_ADDC), 8, "v_addc_co_u32 (vcc in/out) dependent chain") \
    X(a_mbcntlo, I8(F_MBCNTLO), 8, "v_mbcnt_lo alone") \
    X(m_mbcntlo, AB8(F_MUL, F_MBCNTLO), 8, "v_mbcnt_lo | v_mul_f32") \
    X(c_mbcntlo, AB8(F_CVTFU, F_MBCNTLO), 8, "v_mbcnt_lo | v_cvt_f32_u32") \
    X(d_mbcntlo, D8(F_MBCNTLO), 8, "v_mbcnt_lo dependent chain") \
    X(a_lshladd, I8(F_LSHLADD), 8, "v_lshl_add_u32 alone") \
    X(m_lshladd, AB8(F_MUL, F_LSHLADD), 8, "v_lshl_add_u32 | v_mul_f32") \
    X(c_lshladd, AB8(F_CVTFU, F_LSHLADD), 8, "v_lshl_add_u32 | v_cvt_f32_u32") \
    X(d_lshladd, D8(F_LSHLADD), 8, "v_lshl_add_u32 dependent chain") \
    X(a_min3u, I8(F_MIN3U), 8, "v_min3_u32 alone") \
    X(m_min3u, AB8(F_MUL, F_MIN3U), 8, "v_min3_u32 | v_mul_f32") \
    X(c_min3u, AB8(F_CVTFU, F_MIN3U), 8, "v_min3_u32 | v_cvt_f32_u32") \
    X(d_min3u, D8(F_MIN3U), 8, "v_min3_u32 dependent chain") \
    X(a_add3, I8(F_ADD3), 8, "v_add3_u32 alone") \
    X(m_add3, AB8(F_MUL, F_ADD3), 8, "v_add3_u32 | v_mul_f32") \
    X(c_add3, AB8(F_CVTFU, F_ADD3), 8, "v_add3_u32 | v_cvt_f32_u32") \
    X(d_add3, D8(F_ADD3), 8, "v_add3_u32 dependent chain") \
    X(a_bfe, I8(F_BFE), 8, "v_bfe_u32 alone") \
    X(m_bfe, AB8(F_MUL, F_BFE), 8, "v_bfe_u32 | v_mul_f32") \
    X(c_bfe, AB8(F_CVTFU, F_BFE), 8, "v_bfe_u32 | v_cvt_f32_u32") \
    X(d_bfe, D8(F_BFE), 8, "v_bfe_u32 dependent chain") \
    X(a_perm, I8(F_PERM), 8, "v_perm_b32 alone") \
    X(m_perm, AB8(F_MUL, F_PERM), 8, "v_perm_b32 | v_mul_f32") \
    X(c_perm, AB8(F_CVTFU, F_PERM), 8, "v_perm_b32 | v_cvt_f32_u32") \
    X(d_perm, D8(F_PERM), 8, "v_perm_b32 dependent chain") \
    X(a_xad, I8(F_XAD), 8, "v_xad_u32 alone") \
    X(m_xad, AB8(F_MUL, F_XAD), 8, "v_xad_u32 | v_mul_f32") \
    X(c_xad, AB8(F_CVTFU, F_XAD), 8, "v_xad_u32 | v_cvt_f32_u32") \
    X(d_xad, D8(F_XAD), 8, "v_xad_u32 dependent chain") \
    X(a_muls, I8(F_MULS), 8, "v_mul_f32 SGPR src alone") \
    X(m_muls, AB8(F_MUL, F_MULS), 8, "v_mul_f32 SGPR src | v_mul_f32") \
    X(c_muls, AB8(F_CVTFU, F_MULS), 8, "v_mul_f32 SGPR src | v_cvt_f32_u32") \
    X(d_muls, D8(F_MULS), 8, "v_mul_f32 SGPR src dependent chain") \
    X(a_addus, I8(F_ADDUS), 8, "v_add_u32 SGPR src alone") \
    X(m_addus, AB8(F_MUL, F_ADDUS), 8, "v_add_u32 SGPR src | v_mul_f32") \
    X(c_addus, AB8(F_CVTFU, F_ADDUS), 8, "v_add_u32 SGPR src | v_cvt_f32_u32") \
    X(d_addus, D8(F_ADDUS), 8, "v_add_u32 SGPR src dependent chain") \
    X(a_dpp, I8(F_DPP), 8, "v_mov_b32 dpp row_shr alone") \
    X(m_dpp, AB8(F_MUL, F_DPP), 8, "v_mov_b32 dpp row_shr | v_mul_f32") \
    X(c_dpp, AB8(F_CVTFU, F_DPP), 8, "v_mov_b32 dpp row_shr | v_cvt_f32_u32") \
    X(d_dpp, D8(F_DPP), 8, "v_mov_b32 dpp row_shr dependent chain") \
    X(a_sdwa, I8(F_SDWA), 8, "v_and_b32 sdwa alone") \
    X(m_sdwa, AB8(F_MUL, F_SDWA), 8, "v_and_b32 sdwa | v_mul_f32") \
    X(c_sdwa, AB8(F_CVTFU, F_SDWA), 8, "v_and_b32 sdwa | v_cvt_f32_u32") \
    X(d_sdwa, D8(F_SDWA), 8, "v_and_b32 sdwa dependent chain") \
    X(a_readlane, I8(F_READLANE), 8, "v_readlane_b32 alone") \
    X(m_readlane, AB8(F_MUL, F_READLANE), 8, "v_readlane_b32 | v_mul_f32") \
    X(c_readlane, AB8(F_CVTFU, F_READLANE), 8, "v_readlane_b32 | v_cvt_f32_u32") \
    X(a_sadd, I8(F_SADD), 8, "s_add_u32 alone") \
    X(m_sadd, AB8(F_MUL, F_SADD), 8, "s_add_u32 | v_mul_f32") \
    X(c_sadd, AB8(F_CVTFU, F_SADD), 8, "s_add_u32 | v_cvt_f32_u32") \
    X(a_mulcl, I8(F_MULCL), 8, "v_mul_f32 clamp (VOP3) alone") \
    X(c_mulcl, AB8(F_CVTFU, F_MULCL), 8, "v_mul_f32 clamp | v_cvt_f32_u32") \
    X(a_fmaabs, I8(F_FMAABS), 8, "v_fma_f32 |a|,b,1.0 clamp alone") \
    X(c_fmaabs, AB8(F_CVTFU, F_FMAABS), 8, "v_fma_f32 |a|,b,1.0 clamp | v_cvt_f32_u32") \
    X(a_addabs, I8(F_ADDABS), 8, "v_add_f32 |a|,b (VOP3) alone") \
    X(c_addabs, AB8(F_CVTFU, F_ADDABS), 8, "v_add_f32 |a|,b | v_cvt_f32_u32") \
    X(a_cmpabs, I8(F_CMPABS), 8, "v_cmp_lt_f32 |a|,b -> SGPR alone") \
    X(m_cmpabs, AB8(F_MUL, F_CMPABS), 8, "v_cmp_lt_f32 |a|,b -> SGPR | v_mul_f32") \
    X(a_min3f, I8(F_MIN3F), 8, "v_min3_f32 alone") \
    X(m_min3f, AB8(F_MUL, F_MIN3F), 8, "v_min3_f32 | v_mul_f32") \
    X(x_6f2c, F_MUL(0) F_FMAAK(1) F_CVTFU(2) F_ADD(3) F_MUL(4) F_FMAAK(5) F_CMPS(6) F_ADD(7), 8, "mix: 6 float (mul/fmaak/add) + cvt + cmp") \
    X(x_4f4c, F_MUL(0) F_CVTFU(1) F_FMAAK(2) F_CMPS(3) F_ADD(4) F_CNDS(5) F_MUL(6) F_MBCNTLO(7), 8, "mix: 4 float + cvt + cmp + cndmask + mbcnt") \
    X(x_4f4i, F_MUL(0) F_XOR(1) F_FMAAK(2) F_ADDU(3) F_ADD(4) F_LSHR(5) F_MUL(6) F_SUBU(7), 8, "mix: 4 float + 4 simple int") \
    X(x_4f2i2c, F_MUL(0) F_XOR(1) F_FMAAK(2) F_CVTFU(3) F_ADD(4) F_ADDU(5) F_MUL(6) F_CMPS(7), 8, "mix: 4 float + 2 simple int + cvt + cmp") \
    X(x_2f6c, F_MUL(0) F_CVTFU(1) F_CMPS(2) F_CNDS(3) F_ADD(4) F_MBCNTLO(5) F_LSHLADD(6) F_CVTUF(7), 8, "mix: 2 float + 6 complex") \
    X(x_4i4c, F_XOR(0) F_CVTFU(1) F_ADDU(2) F_CMPS(3) F_LSHR(4) F_CNDS(5) F_SUBU(6) F_MBCNTLO(7), 8, "mix: 4 simple int + 4 complex") \
    X(x_philox, F_MAD64(0) F_MAD64(1) F_BITOP3S(2) F_BITOP3S(3) F_MAD64(4) F_MAD64(5) F_BITOP3S(6) F_BITOP3S(7), 8, "mix: Philox rounds (2 mad_u64 + 2 bitop3) x2") \
    X(x_philox_f, F_MAD64(0) F_MUL(1) F_BITOP3S(2) F_FMAAK(3) F_MAD64(4) F_ADD(5) F_BITOP3S(6) F_MUL(7), 8, "mix: Philox ops alternating with float ops") \
    X(x_trans_f, F_LOG(0) F_MUL(1) F_FMAAK(2) F_ADD(3) F_EXP(4) F_MUL(5) F_FMAAK(6) F_ADD(7), 8, "mix: 2 transcendental + 6 float") \
    X(x_trans_c, F_LOG(0) F_CVTFU(1) F_CMPS(2) F_CNDS(3) F_EXP(4) F_CVTUF(5) F_MBCNTLO(6) F_LSHLADD(7), 8, "mix: 2 transcendental + 6 complex") \
    X(z_nop, I8(F_NOP), 8, "s_nop 0")

#define DEF(NAME, BODY, N, DESC) KERNEL(NAME, BODY)
LIST(DEF)

typedef void (*kern_t)(uint32_t*, int, uint32_t, Stamp*);
struct Entry { const char* name; kern_t fn; int per_group; const char* desc; };
#define ENT(NAME, BODY, N, DESC) {#NAME, k_##NAME, N, DESC},
static Entry entries[] = { LIST(ENT) };

int main(int argc, char** argv)
{
    const int tpb = argc > 1 ? atoi(argv[1]) : 1024;
    const char* only = argc > 2 ? argv[2] : nullptr;
    const int blocks = 256, iters = 2000, waves = blocks * tpb / 64;
    printf("# %d threads per block, one block per CU: %d waves per SIMD; %d groups of 8 statements per wave\n", tpb, tpb / 256, iters);
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
        const double instr = 8.0 * iters * e.per_group, wps = tpb / 256.0;
        // whole kernel: first loop start to last loop end, in shader cycles at the measured clock
        const double kern_cyc = (double)(tmax - tmin) / 100.0 * mhz;
        printf("  %-58s %8.3f | %7.2f %7.2f %7.2f | %7.2f | %6.0f %6.2f\n", e.desc, ms,
               cyc.front() / instr / wps, cyc[waves / 2] / instr / wps, cyc.back() / instr / wps,
               kern_cyc / instr / wps, mhz, cyc.back() / cyc.front());
    }
    return 0;
}
