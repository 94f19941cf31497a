#!/usr/bin/env python3
"""
Timing-only variants of the K3 streaming kernel, built from a patched COPY of prosstt_amd/csrc
(the shipped sources carry no experiment switches).  Each variant is a list of (old, new) text
substitutions; outputs of these builds are wrong by construction, only their kernel time is read:

    python tools/ablate.py [names...]          # builds build/ab/libprosstt_amd_<name>.so
    PROSSTT_AMD_LIB=build/ab/libprosstt_amd_s1.so python tools/kbench.py C3

Keeping the cut stages' inputs alive (asm volatile sinks) stops the compiler from deleting the
work that feeds them.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "prosstt_amd", "csrc")
OUT = os.path.join(ROOT, "build", "ab")

RUN_23 = """            while (s1_at - s1_lds >= 64u * 16u) {
                stage2_pass(std::true_type{});
                while (s2_top >= kS2Run) stage3_pass();
            }
        }
        cur = nxt;"""
S1_ONLY = (RUN_23, "            s1_at = s1_lds;\n        }\n        cur = nxt;")

PHILOX_OFF = ("        const prnb::Words W = philox_count_row(ph, quad_hi, quad_lo, k0, k1);",
              "        prnb::Words W; W.w[0] = (ph[0] * 2654435761u) ^ ((uint32_t)g0 * 40503u); W.w[1] = W.w[0] * 3u + k0;\n"
              "        W.w[2] = W.w[1] ^ 0x9E3779B9u; W.w[3] = W.w[2] + W.w[0];")
STORE_2 = """            if (BIG) __builtin_amdgcn_raw_buffer_store_b128(row4, out_rsrc, store_voff, flush_off, 18 /* nt | sc1 */);
            else __builtin_amdgcn_raw_buffer_store_b128(row4, out_rsrc, store_voff, flush_off, 2 /* nt */);"""
STORE_OFF = (STORE_2, STORE_2.replace("store_voff", "0x80000000u"))   # every lane out of range: dropped


SVGPR = ("        const float m4[4] = {cur.x * s, cur.y * s, cur.z * s, cur.w * s};",
         "        float sv_; asm volatile(\"v_mov_b32 %0, %1\" : \"=v\"(sv_) : \"s\"(s));\n"
         "        const float m4[4] = {cur.x * sv_, cur.y * sv_, cur.z * sv_, cur.w * sv_};")
THETA2 = ("            e[j].y = __float_as_uint(PRNB_FMA(a[j], m, bm1[j]));", "            e[j].y = __float_as_uint(a[j] * m + bm1[j]);")

VARIANTS = {
    "base": [],
    # stage 1 only: survivors are pushed, then dropped
    "s1": [S1_ONLY],
    # stages 1 + 2: what stage 2 pushes on S2 is dropped
    "s12": [(RUN_23, "            while (s1_at - s1_lds >= 64u * 16u) { stage2_pass(std::true_type{}); s2_top = 0; }\n        }\n        cur = nxt;")],
    # stage 1 without the Philox call (a 2-instruction hash stands in)
    "s1_nophilox": [S1_ONLY, PHILOX_OFF],
    # stage 1 only, rows not stored (pure issue time of stage 1)
    "s1_nostore": [S1_ONLY, STORE_OFF],
    "s1_nostore_nophilox": [S1_ONLY, STORE_OFF, PHILOX_OFF],
    # everything, mean segments not loaded (constant means)
    "noload": [("            const u32x4_ raw = __builtin_amdgcn_raw_buffer_load_b128(rs, gload_b, 0, 0);",
                "            const u32x4_ raw = {0x3ecccccdu, 0x3f8ccccdu, 0x3d4ccccdu, 0x40200000u}; asm volatile(\"\" :: \"s\"(rs), \"v\"(gload_b));")],
    # everything, rows not stored
    "nostore": [STORE_OFF],
    # K3h (round 6 sources) without the handed-over walks / without the gamma-Poisson samples / with an empty list
    "k3h_nowalk": [("                walk_push(slot < cur.n_walks,", "                walk_push(false && slot < cur.n_walks,")],
    "k3h_noheavy": [("        const unsigned long long mh = __builtin_amdgcn_ballot_w64(has);\n        if (has) {\n            HGEntry h;", "        has = false;\n        const unsigned long long mh = __builtin_amdgcn_ballot_w64(has);\n        if (has) {\n            HGEntry h;")],
    "k3h_none": [("                walk_push(slot < cur.n_walks,", "                walk_push(false && slot < cur.n_walks,"),
                 ("        const unsigned long long mh = __builtin_amdgcn_ballot_w64(has);\n        if (has) {\n            HGEntry h;", "        has = false;\n        const unsigned long long mh = __builtin_amdgcn_ballot_w64(has);\n        if (has) {\n            HGEntry h;")],
    # K3h leaves at once / after its 1/k table / without looking at the list: what a launch of it costs at least
    "k3h_exit0": [("    __shared__ HeavyLds lds_all[kHeavyBlock / 64];\n    const int tid = threadIdx.x,", "    __shared__ HeavyLds lds_all[kHeavyBlock / 64];\n    if (N >= 0) return;\n    const int tid = threadIdx.x,")],
    "k3h_exit1": [("    int hg_top = 0, hp_top = 0, hw_top = 0;      // wave-uniform", "    if (N >= 0) return;\n    int hg_top = 0, hp_top = 0, hw_top = 0;      // wave-uniform")],
    "k3h_onestep": [("base += 4ull * waves) {", "base += 1ull << 40) {")],
    # where K3h's time goes when its list is empty: leave after the tables / skip phase 1 / skip phase 2
    "k3h_exitA": [("    // blk / groups for blk < 2^32:", "    if (N >= 0) return;\n    // blk / groups for blk < 2^32:")],
    "k3h_none_nop1": [("                walk_push(slot < cur.n_walks,", "                walk_push(false && slot < cur.n_walks,"),
                 ("        const unsigned long long mh = __builtin_amdgcn_ballot_w64(has);\n        if (has) {\n            HGEntry h;", "        has = false;\n        const unsigned long long mh = __builtin_amdgcn_ballot_w64(has);\n        if (has) {\n            HGEntry h;"),
                 ("        for (uint64_t r0 = wave_id; r0 < (uint64_t)regions; r0 += 4ull * waves) {", "        for (uint64_t r0 = wave_id; r0 < (uint64_t)regions && N < 0; r0 += 4ull * waves) {")],
    "k3h_none_nop2": [("                walk_push(slot < cur.n_walks,", "                walk_push(false && slot < cur.n_walks,"),
                 ("        const unsigned long long mh = __builtin_amdgcn_ballot_w64(has);\n        if (has) {\n            HGEntry h;", "        has = false;\n        const unsigned long long mh = __builtin_amdgcn_ballot_w64(has);\n        if (has) {\n            HGEntry h;"),
                 ("        for (uint32_t p0 = 0u; p0 < pairs; p0 += 64u) {", "        for (uint32_t p0 = 0u; p0 < pairs && N < 0; p0 += 64u) {")],
    "k3h_none_nop12": [("                walk_push(slot < cur.n_walks,", "                walk_push(false && slot < cur.n_walks,"),
                 ("        const unsigned long long mh = __builtin_amdgcn_ballot_w64(has);\n        if (has) {\n            HGEntry h;", "        has = false;\n        const unsigned long long mh = __builtin_amdgcn_ballot_w64(has);\n        if (has) {\n            HGEntry h;"),
                 ("        for (uint64_t r0 = wave_id; r0 < (uint64_t)regions; r0 += 4ull * waves) {", "        for (uint64_t r0 = wave_id; r0 < (uint64_t)regions && N < 0; r0 += 4ull * waves) {"),
                 ("        for (uint32_t p0 = 0u; p0 < pairs; p0 += 64u) {", "        for (uint32_t p0 = 0u; p0 < pairs && N < 0; p0 += 64u) {")],
    "k3h_exitB": [("    const int32_t groups = (strips + 3) / 4;\n    const uint32_t waves = gridDim.x * (kHeavyBlock / 64);", "    if (N >= 0) return;\n    const int32_t groups = (strips + 3) / 4;\n    const uint32_t waves = gridDim.x * (kHeavyBlock / 64);")],
    "k3h_exitC": [("                walk_push(slot < cur.n_walks,", "                walk_push(false && slot < cur.n_walks,"),
                 ("        const unsigned long long mh = __builtin_amdgcn_ballot_w64(has);\n        if (has) {\n            HGEntry h;", "        has = false;\n        const unsigned long long mh = __builtin_amdgcn_ballot_w64(has);\n        if (has) {\n            HGEntry h;"), ("    // The walks: to one wave in four", "    if (N >= 0) return;\n    // The walks: to one wave in four")],
    "k3h_exitD": [("                walk_push(slot < cur.n_walks,", "                walk_push(false && slot < cur.n_walks,"),
                 ("        const unsigned long long mh = __builtin_amdgcn_ballot_w64(has);\n        if (has) {\n            HGEntry h;", "        has = false;\n        const unsigned long long mh = __builtin_amdgcn_ballot_w64(has);\n        if (has) {\n            HGEntry h;"), ("    // Regions whose list or segment was too small:", "    if (N >= 0) return;\n    // Regions whose list or segment was too small:")],
    # ... the gamma-Poisson entries dropped after the gamma pass
    "k3h_nopois": [("        while (hp_top >= 64) poisson_pass();\n    };", "        hp_top = 0;\n    };")],
    "k3h_gamma1": [("                    ok = last;\n                    if (!ok) {", "                    ok = true;\n                    if (!ok) {")],
    "k3h_pois1": [("                    again = j + 1 < 2 * prnb::kMaxTries;", "                    again = false;")],
    "k3h_grid1024": [("    int heavy_grid = 1536;", "    int heavy_grid = 1024;")],
    "k3h_grid1280": [("    int heavy_grid = 1536;", "    int heavy_grid = 1280;")],
    "k3h_grid2560": [("    int heavy_grid = 1536;", "    int heavy_grid = 2560;")],
    "k3h_grid768": [("    int heavy_grid = 1536;", "    int heavy_grid = 768;")],
    # K3h launched twice (idempotent): what a launch costs when its code and data are warm
    "k3h_twice": [("    k3::sample_counts_heavy_kernel<<<dim3(heavy_blocks),", "    for (int rep_ = 0; rep_ < 2; ++rep_) k3::sample_counts_heavy_kernel<<<dim3(heavy_blocks),")],
    # K3h with a device printf of per-phase cycle counts of a few waves (diagnosis; printf costs registers and time)
    "k3h_trace": [("    int hg_top = 0, hp_top = 0, hw_top = 0;      // wave-uniform",
                   "    int hg_top = 0, hp_top = 0, hw_top = 0;      // wave-uniform\n    int np_l = 0, np_w = 0, np_g = 0, np_p = 0, n_ent = 0; const long long W0 = wall_clock64(); long long T0 = clock64(), TL = 0, TG = 0, TP = 0;"),
                  ("    auto poisson_pass = [&]() __attribute__((always_inline)) {\n", "    auto poisson_pass = [&]() __attribute__((always_inline)) {\n        ++np_p; const long long tp0 = clock64();\n"),
                  ("        hp_top += __popcll(m);\n    };", "        hp_top += __popcll(m);\n        TP += clock64() - tp0;\n    };"),
                  ("    auto gamma_pass = [&]() __attribute__((always_inline)) {\n", "    auto gamma_pass = [&]() __attribute__((always_inline)) {\n        ++np_g; const long long tg0 = clock64();\n"),
                  ("        hp_top += __popcll(mp);\n        while (hp_top >= 64) poisson_pass();", "        hp_top += __popcll(mp);\n        TG += clock64() - tg0;\n        while (hp_top >= 64) poisson_pass();"),
                  ("    auto walk_service = [&](bool drain) __attribute__((always_inline)) {\n        for (;;) {", "    auto walk_service = [&](bool drain) __attribute__((always_inline)) {\n        const long long tl0 = clock64();\n        for (;;) {"),
                  ("            if (busy <= 32 && (drain ? hw_top > 0 : hw_top >= 32)) walk_take();\n            else if (drain ? busy > 0 : busy > 32) walk_pass();\n            else break;\n        }",
                   "            if (busy <= 32 && (drain ? hw_top > 0 : hw_top >= 32)) { walk_take(); ++np_l; }\n            else if (drain ? busy > 0 : busy > 32) { walk_pass(); ++np_w; }\n            else break;\n        }\n        TL += clock64() - tl0;"),
                  ("        hg_top += __popcll(mh);\n        while (hg_top >= 64) gamma_pass();\n    };", "        hg_top += __popcll(mh);\n        n_ent += __popcll(mh);\n        while (hg_top >= 64) gamma_pass();\n    };"),
                  ("    // Regions whose list or segment was too small:", "    const long long T1 = clock64();\n    // Regions whose list or segment was too small:"),
                  ("    walk_service(true);\n    while (hg_top > 0) gamma_pass();\n    while (hp_top > 0) poisson_pass();",
                   "    const long long T2 = clock64();\n    walk_service(true);\n    const long long T3 = clock64();\n    while (hg_top > 0) gamma_pass();\n    while (hp_top > 0) poisson_pass();\n"
                   "    const long long T4 = clock64(); const long long W1 = wall_clock64();\n"
                   "    if (lane == 0 && ((blockIdx.x * 4 + wv) % 61) == 5) printf(\"K3HTRACE wave %d walker %d wall %lld..%lld | ticks total %lld lists %lld ovf %lld walkdrain %lld gpdrain %lld | entries %d | walk %d takes %d passes %lld | gamma %d passes %lld | poisson %d passes %lld\\n\", (int)(blockIdx.x * 4 + wv), (int)walker, W0, W1, T4 - T0, T1 - T0, T2 - T1, T3 - T2, T4 - T3, n_ent, np_l, np_w, TL, np_g, TG, np_p, TP);")],
    # the library default of 10 Philox rounds instead of 7 (timing only: the model is not changed along)
    "philox10": [("constexpr int kCountRounds = 7;", "constexpr int kCountRounds = 10;")],
    # real variants (correct results): tuning constants
    "run16": [("constexpr int kS2Run = 32;", "constexpr int kS2Run = 16;")],
    # the strip-end drain runs to the last walk (nothing is left to K3h there)
    "nobail": [("constexpr int kBail = 6;", "constexpr int kBail = 0;")],
    "bail10": [("constexpr int kBail = 6;", "constexpr int kBail = 10;")],
    "bail16": [("constexpr int kBail = 6;", "constexpr int kBail = 16;")],
    "bail3": [("constexpr int kBail = 6;", "constexpr int kBail = 3;")],
    "bail24": [("constexpr int kBail = 6;", "constexpr int kBail = 24;")],
    "bail32": [("constexpr int kBail = 6;", "constexpr int kBail = 32;")],
    "strip128": [("    g.strip_cells = k3::kStripCells / 2;", "    g.strip_cells = k3::kStripCells;")],
    # small problems (C2): a fixed strip length instead of halving down to 8 (real variants: any strip length is correct)
    "smallstrip10": [("    while (g.strip_cells > 8 && ((n + g.strip_cells - 1) / g.strip_cells) * g.tiles_g < 4 * 5 * 1024) g.strip_cells /= 2;",
                     "    if (((n + g.strip_cells - 1) / g.strip_cells) * g.tiles_g < 4 * 5 * 1024) g.strip_cells = 10;")],
    "smallstrip12": [("    while (g.strip_cells > 8 && ((n + g.strip_cells - 1) / g.strip_cells) * g.tiles_g < 4 * 5 * 1024) g.strip_cells /= 2;",
                     "    if (((n + g.strip_cells - 1) / g.strip_cells) * g.tiles_g < 4 * 5 * 1024) g.strip_cells = 12;")],
    "smallstrip16": [("    while (g.strip_cells > 8 && ((n + g.strip_cells - 1) / g.strip_cells) * g.tiles_g < 4 * 5 * 1024) g.strip_cells /= 2;",
                     "    if (((n + g.strip_cells - 1) / g.strip_cells) * g.tiles_g < 4 * 5 * 1024) g.strip_cells = 16;")],
    "smallstrip20": [("    while (g.strip_cells > 8 && ((n + g.strip_cells - 1) / g.strip_cells) * g.tiles_g < 4 * 5 * 1024) g.strip_cells /= 2;",
                     "    if (((n + g.strip_cells - 1) / g.strip_cells) * g.tiles_g < 4 * 5 * 1024) g.strip_cells = 20;")],
    "smallstrip32": [("    while (g.strip_cells > 8 && ((n + g.strip_cells - 1) / g.strip_cells) * g.tiles_g < 4 * 5 * 1024) g.strip_cells /= 2;",
                     "    if (((n + g.strip_cells - 1) / g.strip_cells) * g.tiles_g < 4 * 5 * 1024) g.strip_cells = 32;")],

    "strip32": [("    g.strip_cells = k3::kStripCells / 2;", "    g.strip_cells = k3::kStripCells / 4;")],
    # plain instead of non-temporal row stores
    "plainstore": [(STORE_2, "            __builtin_amdgcn_raw_buffer_store_b128(row4, out_rsrc, store_voff, flush_off, 0);")],
    # stage 3 waits for 48 / 40 entries (a deeper S2; the block's LDS still allows five per CU)
    "run48": [("constexpr int kS2Run = 32;", "constexpr int kS2Run = 48;"), ("constexpr int kS2Cap = 96;", "constexpr int kS2Cap = 112;")],
    "run40": [("constexpr int kS2Run = 32;", "constexpr int kS2Run = 40;"), ("constexpr int kS2Cap = 96;", "constexpr int kS2Cap = 104;")],
    # round 4: K3h on a second stream with no dependency on the stream kernel (it reads the PREVIOUS call's list: timing only) --
    # the upper bound of what overlapping the two kernels can give
    "k3h_overlap": [
        ("    k3::sample_counts_heavy_kernel<<<dim3(heavy_blocks), dim3(k3::kHeavyBlock), 0, c->stream>>>(",
         "    static hipStream_t s2 = nullptr; static hipEvent_t e2 = nullptr;\n"
         "    if (!s2) { HIP_TRY(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); HIP_TRY(hipEventCreateWithFlags(&e2, hipEventDisableTiming)); }\n"
         "    k3::sample_counts_heavy_kernel<<<dim3(heavy_blocks), dim3(k3::kHeavyBlock), 0, s2>>>("),
        ("    c->list = heavy.list;\n", "    HIP_TRY(hipEventRecord(e2, s2)); HIP_TRY(hipStreamWaitEvent(c->stream, e2, 0));\n    c->list = heavy.list;\n"),
    ],
    # round 4 (real variant): the next cell's record loaded mid-pass into the registers this pass has finished with (no scalar rotation)
    "inplace_scalars": [
        ("""        const float s_next = cinfo[1].s;
        const uint32_t ph_next[4] = {cinfo[1].ph[0], cinfo[1].ph[1], cinfo[1].ph[2], cinfo[1].ph[3]};
        ++cinfo;
        __builtin_amdgcn_sched_barrier(0);
        const prnb::Words W = philox_count_row(ph, quad_hi, quad_lo, k0, k1);
""", """        __builtin_amdgcn_sched_barrier(0);
        const prnb::Words W = philox_count_row(ph, quad_hi, quad_lo, k0, k1);
        __builtin_amdgcn_sched_barrier(0);
        s = cinfo[1].s;
        ph[0] = cinfo[1].ph[0]; ph[1] = cinfo[1].ph[1]; ph[2] = cinfo[1].ph[2]; ph[3] = cinfo[1].ph[3];
        ++cinfo;
        __builtin_amdgcn_sched_barrier(0);
"""),
        ("""        s = s_next;
        ph[0] = ph_next[0]; ph[1] = ph_next[1]; ph[2] = ph_next[2]; ph[3] = ph_next[3];
""", ""),
    ],
    # round 4: the shipped kernel held at four / three blocks per CU by LDS padding (what the fifth block is worth)
    "occ4": [("    __shared__ __attribute__((aligned(kRing * 256))) uint8_t ring_all[kBlock / 64][kRing * 256];\n\n    const int tid = threadIdx.x;",
              "    __shared__ __attribute__((aligned(kRing * 256))) uint8_t ring_all[kBlock / 64][kRing * 256];\n    __shared__ uint32_t occ_pad[2400];\n    if (N < 0) occ_pad[threadIdx.x] = 1u;\n    asm volatile(\"\" :: \"v\"((uint32_t)(uintptr_t)&occ_pad[0]));\n\n    const int tid = threadIdx.x;")],
    "occ3": [("    __shared__ __attribute__((aligned(kRing * 256))) uint8_t ring_all[kBlock / 64][kRing * 256];\n\n    const int tid = threadIdx.x;",
              "    __shared__ __attribute__((aligned(kRing * 256))) uint8_t ring_all[kBlock / 64][kRing * 256];\n    __shared__ uint32_t occ_pad[5600];\n    if (N < 0) occ_pad[threadIdx.x] = 1u;\n    asm volatile(\"\" :: \"v\"((uint32_t)(uintptr_t)&occ_pad[0]));\n\n    const int tid = threadIdx.x;")],
    # round 4: four rows in the ring instead of eight, at the shipped five blocks per CU (padding keeps the block's LDS)
    "ring4": [("constexpr int kRing = 8; ", "constexpr int kRing = 4; "),
              ("    __shared__ __attribute__((aligned(kRing * 256))) uint8_t ring_all[kBlock / 64][kRing * 256];\n\n    const int tid = threadIdx.x;",
               "    __shared__ __attribute__((aligned(kRing * 256))) uint8_t ring_all[kBlock / 64][kRing * 256];\n    __shared__ uint32_t occ_pad[1000];\n    if (N < 0) occ_pad[threadIdx.x] = 1u;\n    asm volatile(\"\" :: \"v\"((uint32_t)(uintptr_t)&occ_pad[0]));\n\n    const int tid = threadIdx.x;")],
    # round 4 (diagnosis): how many counts miss their row in the ring (printf of a few waves: late deliveries, nonzero deliveries)
    "latecount": [("    int late_top = 0;                                // wave-uniform",
                   "    int late_top = 0;                                // wave-uniform\n    int late_total = 0, deliv_total = 0;"),
                  ("                late_top += cnt;\n", "                late_top += cnt;\n                late_total += cnt;\n"),
                  ("        const unsigned long long ml = ok_m & late_m;\n", "        const unsigned long long ml = ok_m & late_m;\n        deliv_total += __popcll(ok_m);\n"),
                  ("    flush_late();\n    if (__builtin_amdgcn_ballot_w64(hpend != kNoHeavy) != 0ull) flush_heavy();",
                   "    flush_late();\n    if (lane == 0 && (region % 4093u) == 7u) printf(\"LATE region %u cells %d delivered %d late %d listed %u\\n\", region, cells, deliv_total, late_total, h_cnt);\n    if (__builtin_amdgcn_ballot_w64(hpend != kNoHeavy) != 0ull) flush_heavy();")],
    # round 4 (real variants): how many waves a small problem is cut into (strips are halved while there are fewer)
    "minwaves10k": [("* g.tiles_g < 4 * 5 * 1024) g.strip_cells /= 2;", "* g.tiles_g < 2 * 5 * 1024) g.strip_cells /= 2;")],
    "minwaves5k": [("* g.tiles_g < 4 * 5 * 1024) g.strip_cells /= 2;", "* g.tiles_g < 5 * 1024) g.strip_cells /= 2;")],
    "minwaves40k": [("* g.tiles_g < 4 * 5 * 1024) g.strip_cells /= 2;", "* g.tiles_g < 8 * 5 * 1024) g.strip_cells /= 2;"), ("g.strip_cells > 8 &&", "g.strip_cells > 4 &&")],
    # round 4 (real variant): the next cell's mean segment AND record loaded mid-pass into the registers this pass has finished
    # with (no rotation at all: ten instructions fewer per pass, one pass of prefetch distance instead of two)
    "inplace_all": [
        ("""        const Seg nn = load_seg(row2);
        const uint64_t row3 = cinfo[3].row_bytes;
        const float s_next = cinfo[1].s;
        const uint32_t ph_next[4] = {cinfo[1].ph[0], cinfo[1].ph[1], cinfo[1].ph[2], cinfo[1].ph[3]};
        ++cinfo;
        __builtin_amdgcn_sched_barrier(0);
        const prnb::Words W = philox_count_row(ph, quad_hi, quad_lo, k0, k1);
""", """        const prnb::Words W = philox_count_row(ph, quad_hi, quad_lo, k0, k1);
        __builtin_amdgcn_sched_barrier(0);
        cur = load_seg(row2);
        row2 = cinfo[2].row_bytes;
        s = cinfo[1].s;
        ph[0] = cinfo[1].ph[0]; ph[1] = cinfo[1].ph[1]; ph[2] = cinfo[1].ph[2]; ph[3] = cinfo[1].ph[3];
        ++cinfo;
        __builtin_amdgcn_sched_barrier(0);
"""),
        ("""        cur = nxt;
        nxt = nn;
        row2 = row3;
        s = s_next;
        ph[0] = ph_next[0]; ph[1] = ph_next[1]; ph[2] = ph_next[2]; ph[3] = ph_next[3];
""", ""),
        ("    Seg cur = load_seg(cinfo[0].row_bytes), nxt = load_seg(cinfo[1].row_bytes);\n    uint64_t row2 = cinfo[2].row_bytes;",
         "    Seg cur = load_seg(cinfo[0].row_bytes);\n    uint64_t row2 = cinfo[1].row_bytes;"),
    ],
    # round 4 (real variant): logical block index = transpose of the hardware's round-robin over the 8 XCDs (one XCD's blocks
    # cover a contiguous eighth of the gene tiles: a tile's slice of the mean tensor in one L2 instead of eight)
    "xcd_transpose": [
        ("    const int32_t tile_g = blockIdx.x / groups;\n    const int32_t strip = (blockIdx.x - tile_g * groups) * 4 + wv;",
         "    const uint32_t per_xcd = gridDim.x >> 3;\n"
         "    const uint32_t blk = blockIdx.x < (per_xcd << 3) ? (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3) : blockIdx.x;\n"
         "    const int32_t tile_g = (int32_t)blk / groups;\n    const int32_t strip = ((int32_t)blk - tile_g * groups) * 4 + wv;"),
        ("    const uint32_t region = blockIdx.x * 4u + (uint32_t)wv;", "    const uint32_t region = blk * 4u + (uint32_t)wv;"),
    ],
    # round 4 (real variants, for C2): strips of 20 / 24 / 12 cells -- one round of waves on the chip's 5120 wave slots
    "strip20": [("    g.strips = (n + g.strip_cells - 1) / g.strip_cells;", "    g.strip_cells = 20;\n    g.strips = (n + g.strip_cells - 1) / g.strip_cells;")],
    "strip24": [("    g.strips = (n + g.strip_cells - 1) / g.strip_cells;", "    g.strip_cells = 24;\n    g.strips = (n + g.strip_cells - 1) / g.strip_cells;")],
    "strip12": [("    g.strips = (n + g.strip_cells - 1) / g.strip_cells;", "    g.strip_cells = 12;\n    g.strips = (n + g.strip_cells - 1) / g.strip_cells;")],
    # mean segments one cell ahead, requested at the END of a pass (one register rotation; the row store gets a whole pass before anything waits behind it)
    "pf1": [("        const Seg nn = load_seg(row2);\n        const uint64_t row3 = cinfo[3].row_bytes;",
             "        const uint64_t row3 = cinfo[3].row_bytes;"),
            ("        cur = nxt;\n        nxt = nn;\n        row2 = row3;", "        cur = nxt;\n        nxt = load_seg(row2);\n        row2 = row3;")],
    # round 4, the two-pipe hypothesis (binary32 mul/add/fma with <= 2 register operands issue beside everything else):
    # instructions moved from the "everything else" side to the binary32 side, count unchanged or higher
    "svgpr": [SVGPR],
    "theta2": [THETA2],
    # C2 (5 000 x 5 000): strips of a fixed number of cells, whatever the problem size
    "c2strip10": [("    g.strips = (n + g.strip_cells - 1) / g.strip_cells;", "    g.strip_cells = 10;\n    g.strips = (n + g.strip_cells - 1) / g.strip_cells;")],
    "c2strip12": [("    g.strips = (n + g.strip_cells - 1) / g.strip_cells;", "    g.strip_cells = 12;\n    g.strips = (n + g.strip_cells - 1) / g.strip_cells;")],
    "c2strip16": [("    g.strips = (n + g.strip_cells - 1) / g.strip_cells;", "    g.strip_cells = 16;\n    g.strips = (n + g.strip_cells - 1) / g.strip_cells;")],
    "c2strip20": [("    g.strips = (n + g.strip_cells - 1) / g.strip_cells;", "    g.strip_cells = 20;\n    g.strips = (n + g.strip_cells - 1) / g.strip_cells;")],
    "c2strip24": [("    g.strips = (n + g.strip_cells - 1) / g.strip_cells;", "    g.strip_cells = 24;\n    g.strips = (n + g.strip_cells - 1) / g.strip_cells;")],
    "c2strip32": [("    g.strips = (n + g.strip_cells - 1) / g.strip_cells;", "    g.strip_cells = 32;\n    g.strips = (n + g.strip_cells - 1) / g.strip_cells;")],
    "c2strip40": [("    g.strips = (n + g.strip_cells - 1) / g.strip_cells;", "    g.strip_cells = 40;\n    g.strips = (n + g.strip_cells - 1) / g.strip_cells;")],
    # one reciprocal for 1/u1 and 1/theta: r = rcp(u1 * th), 1/u1 = r * th, 1/th = r * u1 (a transcendental less, two multiplies more)
    "rcp1": [("    h.iu = hw_rcp(u1);\n    h.t2 = m * (hw_log2(u1) * hw_rcp(u1 - 1.0f));",
              "    const float th_ = u1 - 1.0f;\n    const float r_ = hw_rcp(u1 * th_);\n    h.iu = r_ * th_;\n    h.t2 = m * (hw_log2(u1) * (r_ * u1));")],
}

# round 4, late (the shipped kernel raises its issue priority to 2 in stages 2 and 3 and stores its rows nt | sc1):
VARIANTS.update({
    "prio_off": [("        if (BIG) __builtin_amdgcn_s_setprio(2);           // (see stage2_pass)\n", ""),
                 ("        if (BIG) __builtin_amdgcn_s_setprio(2);\n        const uint32_t top = s1_at;", "        const uint32_t top = s1_at;")],
    "store_nt_only": [(STORE_2, "            __builtin_amdgcn_raw_buffer_store_b128(row4, out_rsrc, store_voff, flush_off, 2);")],
    "prio_off_nt_only": [("        if (BIG) __builtin_amdgcn_s_setprio(2);           // (see stage2_pass)\n", ""),
                         ("        if (BIG) __builtin_amdgcn_s_setprio(2);\n        const uint32_t top = s1_at;", "        const uint32_t top = s1_at;"),
                         (STORE_2, "            __builtin_amdgcn_raw_buffer_store_b128(row4, out_rsrc, store_voff, flush_off, 2);")],
    "prio1": [("        if (BIG) __builtin_amdgcn_s_setprio(2);           // (see stage2_pass)\n", "        __builtin_amdgcn_s_setprio(1);\n"),
              ("        if (BIG) __builtin_amdgcn_s_setprio(2);\n        const uint32_t top = s1_at;", "        __builtin_amdgcn_s_setprio(1);\n        const uint32_t top = s1_at;")],
    # the pushes of stage 1 at the raised priority as well
    "prio_push": [("#pragma unroll\n        for (int h = 0; h < 2; ++h) {\n            uint32_t t0, t1, c_;", "        __builtin_amdgcn_s_setprio(2);\n#pragma unroll\n        for (int h = 0; h < 2; ++h) {\n            uint32_t t0, t1, c_;"),
                  ("        row2 = row3;\n        s = s_next;", "        __builtin_amdgcn_s_setprio(0);\n        row2 = row3;\n        s = s_next;")],
    # mean loads with other cache policies (0 shipped; 2 = nt; 1 = sc0; 16 = sc1)
    "load_aux2": [("__builtin_amdgcn_raw_buffer_load_b128(rs, gload_b, 0, 0);", "__builtin_amdgcn_raw_buffer_load_b128(rs, gload_b, 0, 2);")],
    "load_aux1": [("__builtin_amdgcn_raw_buffer_load_b128(rs, gload_b, 0, 0);", "__builtin_amdgcn_raw_buffer_load_b128(rs, gload_b, 0, 1);")],
    "load_aux16": [("__builtin_amdgcn_raw_buffer_load_b128(rs, gload_b, 0, 0);", "__builtin_amdgcn_raw_buffer_load_b128(rs, gload_b, 0, 16);")],
})

# round 4, late: the order in which the hardware starts the blocks (a permutation of the logical block index: bit-exact)
_BID = ("    const int32_t tile_g = blockIdx.x / groups;\n    const int32_t strip = (blockIdx.x - tile_g * groups) * 4 + wv;",
        "    const int32_t tile_g = bid_ / groups;\n    const int32_t strip = (bid_ - tile_g * groups) * 4 + wv;")
_REG = ("    const uint32_t region = blockIdx.x * 4u + (uint32_t)wv;", "    const uint32_t region = (uint32_t)bid_ * 4u + (uint32_t)wv;")
def _order(expr):
    return [("    const int32_t groups = (strips + 3) / 4;\n    const int32_t tile_g", "    const int32_t groups = (strips + 3) / 4;\n    const int32_t bid_ = %s;\n    const int32_t tile_g" % expr), _BID, _REG]
VARIANTS.update({
    "order_reverse": _order("(int32_t)(gridDim.x - 1u - blockIdx.x)"),
    # strip-major: consecutive blocks take the same strips of consecutive gene tiles
    "order_stripmajor": _order("(int32_t)((blockIdx.x % (gridDim.x / groups)) * groups + blockIdx.x / (gridDim.x / groups))"),
})

# round 4, late: issue priority inside K3h
VARIANTS.update({
    "k3h_prio_walk": [("        const bool busy = wk >= 0;\n        const float* tab", "        __builtin_amdgcn_s_setprio(2);\n        const bool busy = wk >= 0;\n        const float* tab"),
                      ("            wrem = b4;\n        }\n    };", "            wrem = b4;\n        }\n        __builtin_amdgcn_s_setprio(0);\n    };")],
    "k3h_prio_gp": [("        const int cnt = hp_top < 64 ? hp_top : 64;\n        bool again = false, small = false;", "        __builtin_amdgcn_s_setprio(2);\n        const int cnt = hp_top < 64 ? hp_top : 64;\n        bool again = false, small = false;"),
                    ("        hp_top += __popcll(m);\n    };", "        hp_top += __popcll(m);\n        __builtin_amdgcn_s_setprio(0);\n    };"),
                    ("        const int cnt = hg_top < 64 ? hg_top : 64;\n        bool again = false, accepted = false;", "        __builtin_amdgcn_s_setprio(2);\n        const int cnt = hg_top < 64 ? hg_top : 64;\n        bool again = false, accepted = false;"),
                    ("        hp_top += __popcll(mp);\n        while (hp_top >= 64) poisson_pass();", "        hp_top += __popcll(mp);\n        __builtin_amdgcn_s_setprio(0);\n        while (hp_top >= 64) poisson_pass();")],
})

# round 4, late: the two scheduling barriers around stage 1's load block left to the (max-ilp) scheduler
VARIANTS.update({
    "no_sched_barriers": [("        __builtin_amdgcn_sched_barrier(0);\n        const Seg nn = load_seg(row2);", "        const Seg nn = load_seg(row2);"),
                          ("        ++cinfo;\n        __builtin_amdgcn_sched_barrier(0);\n        const prnb::Words W", "        ++cinfo;\n        const prnb::Words W")],
    "no_sched_barrier_2": [("        ++cinfo;\n        __builtin_amdgcn_sched_barrier(0);\n        const prnb::Words W", "        ++cinfo;\n        const prnb::Words W")],
})


# round 5: where does the first CHECKED call of a process spend its time?  (stderr, synchronising: diagnosis only)
_T = 'static auto T0_ = std::chrono::steady_clock::now();\n'
def _tick(label):
    return ('    { (void)hipStreamSynchronize(c->stream); auto t_ = std::chrono::steady_clock::now(); fprintf(stderr, "COLDTRACE %%-28s %%9.3f ms\\n", "%s", '
            'std::chrono::duration<double, std::milli>(t_ - T0_).count()); T0_ = t_; }\n' % label)
VARIANTS["cold_trace"] = [
    ("#define PA_EXPORT extern", "#include <chrono>\n" + _T + "#define PA_EXPORT extern"),
    ("            if ((size_t)rows > c->row_bad_cap) {\n", _tick("before row_bad alloc") + "            if ((size_t)rows > c->row_bad_cap) {\n"),
    ("            row_flags_kernel<<<dim3((unsigned)(rows < 65536 ? rows : 65536))", _tick("row_bad alloc") + "            row_flags_kernel<<<dim3((unsigned)(rows < 65536 ? rows : 65536))"),
    ("    const int64_t span = (N + 4 > G ? N + 4 : G);\n", _tick("row_flags_kernel") + "    const int64_t span = (N + 4 > G ? N + 4 : G);\n"),
    ("    const bool vec = (G % 4 == 0)", _tick("prep_kernel + setup") + "    const bool vec = (G % 4 == 0)"),
    ("    c->list = heavy.list;\n", _tick("stream kernel + K3h") + "    c->list = heavy.list;\n"),
    ("    c->call_parity ^= 1u;\n", _tick("(end of the call)") + "    c->call_parity ^= 1u;\n"),
]


# round 5 (real variants, on cells presented grouped by mean-tensor row): no mean load when the cell two ahead sits on the row
# of the cell one ahead; stage 3 waits for other stack depths (its passes now take eight terms)
VARIANTS.update({
    "skipload": [("        const Seg nn = load_seg(row2);\n", "        Seg nn = nxt;\n        if (row2 != row1) nn = load_seg(row2);\n"),
                 ("        row2 = row3;\n        s = s_next;", "        row1 = row2;\n        row2 = row3;\n        s = s_next;"),
                 ("    uint64_t row2 = cinfo[2].row_bytes;\n", "    uint64_t row2 = cinfo[2].row_bytes, row1 = cinfo[1].row_bytes;\n")],
    "run16b": [("constexpr int kS2Run = 32;", "constexpr int kS2Run = 16;")],
    "run24": [("constexpr int kS2Run = 32;", "constexpr int kS2Run = 24;")],
    "run32c": [("constexpr int kS2Run = 40;", "constexpr int kS2Run = 32;")],
    "run36c": [("constexpr int kS2Run = 40;", "constexpr int kS2Run = 36;")],
    "run28c": [("constexpr int kS2Run = 40;", "constexpr int kS2Run = 28;")],
    "run30c": [("constexpr int kS2Run = 40;", "constexpr int kS2Run = 30;")],
    "run34c": [("constexpr int kS2Run = 40;", "constexpr int kS2Run = 34;")],
    "run24c": [("constexpr int kS2Run = 40;", "constexpr int kS2Run = 24;")],
    "run48c": [("constexpr int kS2Run = 40;", "constexpr int kS2Run = 48;"), ("constexpr int kS2Cap = 104;", "constexpr int kS2Cap = 112;")],
    "bail4": [("constexpr int kBail = 6;", "constexpr int kBail = 4;")],
    "bail8": [("constexpr int kBail = 6;", "constexpr int kBail = 8;")],
    "bail12": [("constexpr int kBail = 6;", "constexpr int kBail = 12;")],
    "run44": [("constexpr int kS2Run = 32;", "constexpr int kS2Run = 44;"), ("constexpr int kS2Cap = 96;", "constexpr int kS2Cap = 108;")],
    "run48b": [("constexpr int kS2Run = 32;", "constexpr int kS2Run = 48;"), ("constexpr int kS2Cap = 96;", "constexpr int kS2Cap = 112;")],
    "run48_bail10": [("constexpr int kS2Run = 32;", "constexpr int kS2Run = 48;"), ("constexpr int kS2Cap = 96;", "constexpr int kS2Cap = 112;"), ("constexpr int kBail = 6;", "constexpr int kBail = 10;")],
    "run40_bail10": [("constexpr int kS2Run = 32;", "constexpr int kS2Run = 40;"), ("constexpr int kS2Cap = 96;", "constexpr int kS2Cap = 104;"), ("constexpr int kBail = 6;", "constexpr int kBail = 10;")],
    "bail10b": [("constexpr int kBail = 6;", "constexpr int kBail = 10;")],
    "bail3b": [("constexpr int kBail = 6;", "constexpr int kBail = 3;")],
})


# round 5 (timing only): what does a scalar / a vector instruction more per pass of stage 1 cost?
_S8 = ' '.join(['"s_add_u32 %0, %0, 1\\n\\t"'] * 8)
_V4 = ' '.join(['"v_add_u32 %0, %0, %0\\n\\t"'] * 4)
VARIANTS.update({
    "salu8": [("        const prnb::Words W = philox_count_row(ph, quad_hi, quad_lo, k0, k1);",
               "        { uint32_t dummy_ = (uint32_t)cl; asm volatile(" + _S8 + " : \"+s\"(dummy_)); }\n        const prnb::Words W = philox_count_row(ph, quad_hi, quad_lo, k0, k1);")],
    "valu4": [("        const prnb::Words W = philox_count_row(ph, quad_hi, quad_lo, k0, k1);",
               "        { uint32_t dummy_ = (uint32_t)lane; asm volatile(" + _V4 + " : \"+v\"(dummy_)); }\n        const prnb::Words W = philox_count_row(ph, quad_hi, quad_lo, k0, k1);")],
})


# round 5 (timing only: another DEFINITION -- the model is not changed along): the inversion class up to theta 24 / 32 instead of 16
VARIANTS.update({
    "light24": [("constexpr float kLightTheta = 16.0f;", "constexpr float kLightTheta = 24.0f;"),
                ("    if (!(__builtin_fmaxf(th_edge, bm1) <= 15.9f)) phi = 3.0e38f;", "    if (!(__builtin_fmaxf(th_edge, bm1) <= 23.9f)) phi = 3.0e38f;")],
    "light32": [("constexpr float kLightTheta = 16.0f;", "constexpr float kLightTheta = 32.0f;"),
                ("    if (!(__builtin_fmaxf(th_edge, bm1) <= 15.9f)) phi = 3.0e38f;", "    if (!(__builtin_fmaxf(th_edge, bm1) <= 31.9f)) phi = 3.0e38f;")],
})


# round 6: the unfinished walks of a strip travel to K3h with their state -- how many may be left when the strip's drain stops
def _bail(n, slots=None):
    return [("constexpr int kBail = 6; ", "constexpr int kBail = %d; " % n)]
VARIANTS.update({"r6_bail0": _bail(0), "r6_bail2": _bail(2), "r6_bail3": _bail(3), "r6_bail8": _bail(8)})


# round 6 (timing only: the walk's groups would have to move in the definition): stage 2 takes the terms k = 0..4, not 0..2
VARIANTS["r6_s2_5terms"] = [
    ("        const unsigned long long hit_m = K3_MASK(r2 < 0.0f);\n        const unsigned long long tail_m = K3_MASK(ps2 < 1.0f);",
     "        const float ps3 = ps2 * PRNB_FMA(dd, 0.33333334f, qq);\n        const float r3 = r2 - ps3;\n        const float ps4 = ps3 * PRNB_FMA(dd, 0.25f, qq);\n        const float r4 = r3 - ps4;\n"
     "        const unsigned long long hit_m = K3_MASK(r4 < 0.0f);\n        const unsigned long long tail_m = K3_MASK(ps4 < 1.0f);"),
    ("        const uint32_t res = (uint32_t)((2 + ((int32_t)prnb::f2u(r0) >> 31)) + ((int32_t)prnb::f2u(r1) >> 31));",
     "        const uint32_t res = (uint32_t)(((4 + ((int32_t)prnb::f2u(r0) >> 31)) + ((int32_t)prnb::f2u(r1) >> 31)) + (((int32_t)prnb::f2u(r2) >> 31) + ((int32_t)prnb::f2u(r3) >> 31)));"),
    ("            e2.x = ps2 * PRNB_FMA(dd, 0.33333334f, qq);      // pmf at k = 3 (the 1/k table's 1/3)", "            e2.x = ps4 * PRNB_FMA(dd, 0.2f, qq);"),
    ("            e2.w = r2;", "            e2.w = r4;"),
]
VARIANTS["r6_s2_4terms"] = [
    ("        const unsigned long long hit_m = K3_MASK(r2 < 0.0f);\n        const unsigned long long tail_m = K3_MASK(ps2 < 1.0f);",
     "        const float ps3 = ps2 * PRNB_FMA(dd, 0.33333334f, qq);\n        const float r3 = r2 - ps3;\n"
     "        const unsigned long long hit_m = K3_MASK(r3 < 0.0f);\n        const unsigned long long tail_m = K3_MASK(ps3 < 1.0f);"),
    ("        const uint32_t res = (uint32_t)((2 + ((int32_t)prnb::f2u(r0) >> 31)) + ((int32_t)prnb::f2u(r1) >> 31));",
     "        const uint32_t res = (uint32_t)(((3 + ((int32_t)prnb::f2u(r0) >> 31)) + ((int32_t)prnb::f2u(r1) >> 31)) + ((int32_t)prnb::f2u(r2) >> 31));"),
    ("            e2.x = ps2 * PRNB_FMA(dd, 0.33333334f, qq);      // pmf at k = 3 (the 1/k table's 1/3)", "            e2.x = ps3 * PRNB_FMA(dd, 0.25f, qq);"),
    ("            e2.w = r2;", "            e2.w = r3;"),
]
VARIANTS["r6_s2_7terms"] = [
    ("        const unsigned long long hit_m = K3_MASK(r2 < 0.0f);\n        const unsigned long long tail_m = K3_MASK(ps2 < 1.0f);",
     "        const float ps3 = ps2 * PRNB_FMA(dd, 0.33333334f, qq);\n        const float r3 = r2 - ps3;\n        const float ps4 = ps3 * PRNB_FMA(dd, 0.25f, qq);\n        const float r4 = r3 - ps4;\n"
     "        const float ps5 = ps4 * PRNB_FMA(dd, 0.2f, qq);\n        const float r5 = r4 - ps5;\n        const float ps6 = ps5 * PRNB_FMA(dd, 0.16666667f, qq);\n        const float r6 = r5 - ps6;\n"
     "        const unsigned long long hit_m = K3_MASK(r6 < 0.0f);\n        const unsigned long long tail_m = K3_MASK(ps6 < 1.0f);"),
    ("        const uint32_t res = (uint32_t)((2 + ((int32_t)prnb::f2u(r0) >> 31)) + ((int32_t)prnb::f2u(r1) >> 31));",
     "        const uint32_t res = (uint32_t)((((6 + ((int32_t)prnb::f2u(r0) >> 31)) + ((int32_t)prnb::f2u(r1) >> 31)) + (((int32_t)prnb::f2u(r2) >> 31) + ((int32_t)prnb::f2u(r3) >> 31))) + (((int32_t)prnb::f2u(r4) >> 31) + ((int32_t)prnb::f2u(r5) >> 31)));"),
    ("            e2.x = ps2 * PRNB_FMA(dd, 0.33333334f, qq);      // pmf at k = 3 (the 1/k table's 1/3)", "            e2.x = ps6 * PRNB_FMA(dd, 0.14285715f, qq);"),
    ("            e2.w = r2;", "            e2.w = r6;"),
]


# round 6 (timing only): what the end of a strip costs the streaming kernel -- no hand-over at all / the atomics but no copy
VARIANTS["r6_end_none"] = [("    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0)\n    const int n_kept", "    if (N >= 0) { for (int cl = (cells > kRing ? cells - kRing : 0); cl < cells; ++cl) flush_row(cl); flush_late(); return; }\n    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0)\n    const int n_kept")]
VARIANTS["r6_end_nowait"] = [("    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0)\n    const int n_kept", "    const int n_kept")]
VARIANTS["r6_end_nocopy"] = [("    for (uint32_t i = (uint32_t)lane; i < n_ent; i += 64u) {\n        const unsigned long long raw = i < 64u ? first_raw : staged(i);", "    for (uint32_t i = (uint32_t)lane; i < n_ent && N < 0; i += 64u) {\n        const unsigned long long raw = i < 64u ? first_raw : staged(i);"),
                              ("    if ((uint32_t)lane < n_ent) first_raw = staged((uint32_t)lane);", "")]


# round 6: how many of K3h's blocks walk (of every 16)
def _walkers(n):
    return [("constexpr uint32_t kWalkerBlocks = 6u;", "constexpr uint32_t kWalkerBlocks = %du;" % n)]
VARIANTS.update({"k3h_walk2": _walkers(2), "k3h_walk4": _walkers(4), "k3h_walk5": _walkers(5), "k3h_walk6": _walkers(6), "k3h_walk8": _walkers(8)})


def build(name):
    work = os.path.join(OUT, "src_" + name)
    shutil.rmtree(work, ignore_errors=True)
    shutil.copytree(SRC, os.path.join(work, "prosstt_amd", "csrc"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(work, "include"))
    files = {fn: open(os.path.join(work, "prosstt_amd", "csrc", fn)).read()
             for fn in ("k3_stream.h", "k3_heavy.h", "prnb_device.h", "prosstt_amd.hip")}
    for old, new in VARIANTS[name]:
        hits = [fn for fn, text in files.items() if old in text]
        if not hits:
            raise SystemExit("variant %s: anchor not found:\n%s" % (name, old))
        for fn in hits:
            files[fn] = files[fn].replace(old, new)
    for fn, text in files.items():
        open(os.path.join(work, "prosstt_amd", "csrc", fn), "w").write(text)
    lib = os.path.join(OUT, "libprosstt_amd_%s.so" % name)
    # ABLATE_FLAGS="-mllvm -amdgpu-sched-strategy=max-memory-clause" ABLATE_SUFFIX=_maxmem: the same variant under other compiler flags
    lib = os.path.join(OUT, "libprosstt_amd_%s%s.so" % (name, os.environ.get("ABLATE_SUFFIX", "")))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
                           "-mllvm", "-amdgpu-sched-strategy=max-ilp",
                           *os.environ.get("ABLATE_FLAGS", "").split(),
                           "-fPIC", "-shared", "-fvisibility=hidden", "-o", lib,
                           os.path.join(work, "prosstt_amd", "csrc", "prosstt_amd.hip")])
    shutil.rmtree(work, ignore_errors=True)
    return lib


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for n in (sys.argv[1:] or list(VARIANTS)):
        print("built", build(n))
