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

RUN_23 = """        while (s1_top >= 64) {
            stage2_pass();
            while (s2_top >= kS2Run) stage3_pass();
        }
        cur = nxt;"""

VARIANTS = {
    "base": [],
    # stage 1 only: survivors are pushed, then dropped
    "s1": [(RUN_23, "        s1_top = 0;\n        cur = nxt;")],
    # stages 1 + 2: what stage 2 pushes on S2 is dropped
    "s12": [(RUN_23, "        while (s1_top >= 64) { stage2_pass(); s2_top = 0; }\n        cur = nxt;")],
    # no late results (they are dropped instead of listed)
    "nolate": [("        const unsigned long long ml = __builtin_amdgcn_ballot_w64(res > 0) & __builtin_amdgcn_ballot_w64(late);",
                "        const unsigned long long ml = 0ull; asm volatile(\"\" :: \"v\"((int)late));")],
    # stage 1 without the Philox call (a 2-instruction hash stands in)
    "s1_nophilox": [(RUN_23, "        s1_top = 0;\n        cur = nxt;"),
                    ("        const prnb::Words W = prnb::philox4x32_10<2>(c_lo, c_hi, (uint32_t)g0 >> 2, 0u, k0, k1);",
                     "        prnb::Words W; W.w[0] = (c_lo * 2654435761u) ^ ((uint32_t)g0 * 40503u); W.w[1] = W.w[0] * 3u + k0;\n"
                     "        W.w[2] = W.w[1] ^ 0x9E3779B9u; W.w[3] = W.w[2] + W.w[0];")],
    # stage 1 without the S1 push
    # stages 1 + 2 without the listing of given-up samples
    "s12_nolist": [(RUN_23, "        while (s1_top >= 64) { stage2_pass(); s2_top = 0; }\n        cur = nxt;"),
                   ("        list_sample(give_m, give_up, p2);", "        asm volatile(\"\" :: \"v\"((int)give_up), \"s\"(give_m));")],
    # stages 1 + 2 with plain multiplies standing in for v_rcp / v_log / v_exp
    # stages 1 + 2 without deliver
    "s12_nodeliver": [(RUN_23, "        while (s1_top >= 64) { stage2_pass(); s2_top = 0; }\n        cur = nxt;"),
                      ("        deliver(p2, res);\n        list_sample(give_m, give_up, p2);", "        asm volatile(\"\" :: \"v\"(res), \"v\"(p2));\n        list_sample(give_m, give_up, p2);")],
    # stage 1 only, rows not stored (pure issue time of stage 1)
    "s1_nostore": [(RUN_23, "        s1_top = 0;\n        cur = nxt;"),
                   ("        if (g0 < G) {\n            if (VEC) {", "        if (g0 < G && v[0] == 12345) {\n            if (VEC) {")],
    "s1_nostore_nophilox": [(RUN_23, "        s1_top = 0;\n        cur = nxt;"),
                   ("        if (g0 < G) {\n            if (VEC) {", "        if (g0 < G && v[0] == 12345) {\n            if (VEC) {"),
                   ("        const prnb::Words W = prnb::philox4x32_10<2>(c_lo, c_hi, (uint32_t)g0 >> 2, 0u, k0, k1);",
                    "        prnb::Words W; W.w[0] = (c_lo * 2654435761u) ^ ((uint32_t)g0 * 40503u); W.w[1] = W.w[0] * 3u + k0;\n"
                    "        W.w[2] = W.w[1] ^ 0x9E3779B9u; W.w[3] = W.w[2] + W.w[0];")],
    # everything, mean segments not loaded (constant means): is the row stores' cost the wait they share with the loads?
    "noload": [("            const float4 v = *reinterpret_cast<const float4*>(rowp + gload);",
                "            const float4 v = make_float4(0.4f, 1.1f, 0.05f, 2.5f); asm volatile(\"\" :: \"v\"(rowp + gload));")],
    # everything, rows not stored
    "nostore": [("        if (g0 < G) {\n            if (VEC) {", "        if (g0 < G && v[0] == 12345) {\n            if (VEC) {")],
    # K3h without the redo walks / without the gamma-Poisson samples
    "k3h_noredo": [("            light = P.valid && P.light;\n", "            light = false;\n")],
    "k3h_noheavy": [("            heavy = P.valid && !P.light;\n", "            heavy = false;\n")],
    "k3h_none": [("            light = P.valid && P.light;\n", "            light = false;\n"), ("            heavy = P.valid && !P.light;\n", "            heavy = false;\n")],
    "k3h_grid1024": [("k3::sample_counts_heavy_kernel<<<dim3(2048),", "k3::sample_counts_heavy_kernel<<<dim3(1024),")],
    "k3h_grid4096": [("k3::sample_counts_heavy_kernel<<<dim3(2048),", "k3::sample_counts_heavy_kernel<<<dim3(4096),")],
    "k3h_grid8192": [("k3::sample_counts_heavy_kernel<<<dim3(2048),", "k3::sample_counts_heavy_kernel<<<dim3(8192),")],
    # occupancy experiment: S1 too small for the worst case (fine on C3 in practice): 5 blocks per CU instead of 4
    # candidate changes of the sampler's definition, timing only (the model is not changed along)
    "philox7": [("    for (int round = 0; round < 10; ++round) {\n        // one 32x32->64 product", "    for (int round = 0; round < 7; ++round) {\n        // one 32x32->64 product")],
    # what the threshold margins cost (no sample is ever given up: wrong in 1e-3 of the walks)
    "s3_nomargin": [("        const uint32_t near = umin(umin(rem1 + dl, rem2 + dl), umin(rem3 + dl, rem4 + dl));\n        const bool close = near < 2u * dl;               // never for an idle lane: its dl is 0\n",
                     "        const bool close = false; const uint32_t near = 0xffffffffu;\n")],
    "s23_nomargin": [("        const uint32_t near = umin(umin(rem1 + dl, rem2 + dl), umin(rem3 + dl, rem4 + dl));\n        const bool close = near < 2u * dl;               // never for an idle lane: its dl is 0\n",
                      "        const bool close = false; const uint32_t near = 0xffffffffu;\n"),
                     ("        const uint32_t near = umin(umin(rem1 + d2, rem2 + d2), rem3 + d2);\n", "        const uint32_t near = 0xffffffffu;\n")],
    # stage 3 without the delivery of results
    "s3_nodeliver": [("        deliver(pos, res);\n        list_sample(close_m, close | big, pos);", "        asm volatile(\"\" :: \"v\"(res), \"v\"(pos));\n        list_sample(close_m, close | big, pos);")],
    # wave priority: stage 1 (memory issue) above stages 2/3, or the other way round
    "prio_s1": [("        const float M[4] = {cur.M[0], cur.M[1], cur.M[2], cur.M[3]};\n", "        __builtin_amdgcn_s_setprio(2);\n        const float M[4] = {cur.M[0], cur.M[1], cur.M[2], cur.M[3]};\n"),
                ("        while (s1_top >= 64) {\n            stage2_pass();\n            while (s2_top >= kS2Run) stage3_pass();\n        }\n        cur = nxt;",
                 "        __builtin_amdgcn_s_setprio(0);\n        while (s1_top >= 64) {\n            stage2_pass();\n            while (s2_top >= kS2Run) stage3_pass();\n        }\n        cur = nxt;")],
    "prio_s23": [("        const float M[4] = {cur.M[0], cur.M[1], cur.M[2], cur.M[3]};\n", "        __builtin_amdgcn_s_setprio(0);\n        const float M[4] = {cur.M[0], cur.M[1], cur.M[2], cur.M[3]};\n"),
                 ("        while (s1_top >= 64) {\n            stage2_pass();\n            while (s2_top >= kS2Run) stage3_pass();\n        }\n        cur = nxt;",
                  "        __builtin_amdgcn_s_setprio(2);\n        while (s1_top >= 64) {\n            stage2_pass();\n            while (s2_top >= kS2Run) stage3_pass();\n        }\n        cur = nxt;")],
    # block order: gene tiles fastest (concurrent blocks write few rows of the count matrix, all their tiles)
    "tilefast": [("    const int32_t tile_g = blockIdx.x / groups;\n    const int32_t strip = (blockIdx.x - tile_g * groups) * 4 + wv;",
                  "    const int32_t tiles_all = (G + kTileG - 1) / kTileG;\n    const int32_t grp_ = blockIdx.x / tiles_all;\n    const int32_t tile_g = blockIdx.x - grp_ * tiles_all;\n    const int32_t strip = grp_ * 4 + wv;"),
                 ("            const int32_t tile_g = blk / groups;\n            const int64_t n0 = (int64_t)((blk - tile_g * groups) * 4 + (int32_t)(r & 3)) * strip_cells;",
                  "            const int32_t tiles_all = (G + kTileG - 1) / kTileG;\n            const int32_t grp_ = blk / tiles_all;\n            const int32_t tile_g = blk - grp_ * tiles_all;\n            const int64_t n0 = (int64_t)(grp_ * 4 + (int32_t)(r & 3)) * strip_cells;")],
    # real variants (correct results): tuning constants
    "run16": [("constexpr int kS2Run = 32;", "constexpr int kS2Run = 16;")],
    "run48": [("constexpr int kS2Run = 32;", "constexpr int kS2Run = 48;"), ("constexpr int kS2Cap = 96;", "constexpr int kS2Cap = 112;")],
    "strip128": [("int64_t strip_cells = k3::kStripCells / 2;", "int64_t strip_cells = k3::kStripCells;")],
    "strip32": [("int64_t strip_cells = k3::kStripCells / 2;", "int64_t strip_cells = k3::kStripCells / 4;")],
    # stage 1 without the push and with a small S1: 6 waves per SIMD instead of 4 (is stage 1 latency-bound?)
    "s1_nopush_occ6": [(RUN_23, "        s1_top = 0;\n        cur = nxt;"),
                       ('            asm volatile("s_mov_b64 exec, %0\\n\\tds_write_b128 %1, %2\\n\\ts_mov_b64 exec, -1"\n'
                        '                         :: "s"(push_m), "v"(slot), "v"(e) : "memory");',
                        '            asm volatile("" :: "s"(push_m), "v"(slot), "v"(e));'),
                       ("constexpr int kS1Cap = 320;", "constexpr int kS1Cap = 16;"),
                       ('static_assert(kS2Cap >= kS2Run - 1 + 64 && kS1Cap >= 63 + 256, "a stack must take one more pass of pushes");', ""),
             # never write beyond the stack (entries are lost instead: wrong counts, valid addresses)
             ("            const uint32_t slot = (s1_lds + ((uint32_t)s1_top << 4)) + ((uint32_t)lane_rank(push_m) << 4);",
              "            const uint32_t slot = s1_lds + (umin((uint32_t)s1_top + (uint32_t)lane_rank(push_m), (uint32_t)kS1Cap - 1u) << 4);"),
             ("            s1_top += __popcll(push_m);", "            s1_top += __popcll(push_m); s1_top = s1_top < kS1Cap ? s1_top : kS1Cap;")],
}


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
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           "-fPIC", "-shared", "-fvisibility=hidden", "-o", lib,
                           os.path.join(work, "prosstt_amd", "csrc", "prosstt_amd.hip")])
    shutil.rmtree(work, ignore_errors=True)
    return lib


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for n in (sys.argv[1:] or list(VARIANTS)):
        print("built", build(n))
