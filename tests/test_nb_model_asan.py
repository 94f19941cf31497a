"""
CPU: the scalar model of the count sampler under AddressSanitizer / UBSan at the corner of the inversion class
(theta = 24, -log P(X = 0) just under 19: means around 141, tail ratio 24/25; PRNB-5's corner at theta = 16 as well), where walks are longest.  Round 3's
definition let a walk run past its 1/k table there (a read beyond the table in the model, beyond the LDS copy on the
device); PRNB-5 / PRNB-6 end every walk at k = 1022, PRNB-7 at k = 1024, and the table covers it.  The driver below is compiled together with
oracle/nb_model.c (no GPU: the libm stand-ins of the hardware functions).  Round 6: the gamma-Poisson class's question / tape /
replay resolver (PRNB-7: nb_model.c, resolve_waiting) runs under the sanitizers as well, answered by the stand-ins THROUGH the query
machinery -- its counts must equal the direct evaluation's, and leak detection is on for that part.
"""
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r"""
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
void prnb_sample_iid(float m, double a, double b, uint64_t seed, uint64_t first_cell, uint32_t gene, int64_t n, int32_t* out);
void prnb_set_hw_query_standins(void);
void prnb_set_hw_query(void* fn);
void prnb_query_stats(int64_t* rounds, int64_t* values);
static int resolver_case(float m, double a, double b, int64_t n)
{
    /* the gamma-Poisson class through the resolver (questions, tapes, replay) == the direct evaluation */
    int32_t* direct = malloc(sizeof(int32_t) * n);
    int32_t* asked = malloc(sizeof(int32_t) * n);
    prnb_sample_iid(m, a, b, 4242, 77, 3, n, direct);
    prnb_set_hw_query_standins();
    prnb_sample_iid(m, a, b, 4242, 77, 3, n, asked);
    int64_t rounds = 0, values = 0;
    prnb_query_stats(&rounds, &values);
    prnb_set_hw_query(0);
    int bad = 0;
    for (int64_t i = 0; i < n; ++i) bad += direct[i] != asked[i];
    printf("resolver m = %g a = %g b = %g: %lld rounds, %lld values, %d differ\n", (double)m, a, b, (long long)rounds, (long long)values, bad);
    free(direct); free(asked);
    return bad != 0 || rounds < 3;
}
int main(void)
{
    /* theta > 24 (PTRS), r < 1 (the boost, small lambda), the Poisson limit at a large mean, a huge mean */
    if (resolver_case(300.0f, 0.3, 2.0, 60000) || resolver_case(8.0f, 3.0, 40.0, 60000) || resolver_case(40.0f, 0.0, 1.00000001, 60000) ||
        resolver_case(3.0e5f, 0.3, 2.0, 5000)) return 5;
    const int64_t n = 400000;
    int32_t* out = malloc(sizeof(int32_t) * n);
    /* (m, alpha, beta): theta = alpha*m + beta - 1 */
    const double cases[][3] = {{101.0, 0.1366, 2.0}, {100.0, 0.14, 2.0}, {107.0, 0.1402, 2.0}, {106.9, 0.0, 17.0},
                               {60.0, 0.25, 2.0}, {18.9, 0.0, 1.00000001}, {90.0, 0.1, 3.0},
                               {141.0, 0.1631, 2.0}, {135.0, 0.163, 2.0}, {141.5, 0.0, 25.0}, {120.0, 0.19, 2.0}};
    long worst = 0;
    for (unsigned c = 0; c < sizeof(cases) / sizeof(cases[0]); ++c) {
        prnb_sample_iid((float)cases[c][0], cases[c][1], cases[c][2], 77 + c, 1000000ull * c, c, n, out);
        double sum = 0;
        for (int64_t i = 0; i < n; ++i) { sum += out[i]; if (out[i] > worst) worst = out[i]; if (out[i] < 0) return 3; }
        printf("case %u: mean %.3f (m = %.1f)\n", c, sum / n, cases[c][0]);
        if (sum / n < 0.97 * cases[c][0] || sum / n > 1.03 * cases[c][0]) return 2;
    }
    printf("largest count %ld\n", worst);
    free(out);
    return worst > 100000 ? 4 : 0;
}
"""


def test_model_walks_stay_inside_their_table_under_asan():
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    tmp = tempfile.mkdtemp(prefix="prnb_asan_")
    try:
        open(os.path.join(tmp, "driver.c"), "w").write(DRIVER)
        exe = os.path.join(tmp, "driver")
        build = subprocess.run([gcc, "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                                "-fvisibility=default", "-o", exe, os.path.join(tmp, "driver.c"),
                                os.path.join(ROOT, "oracle", "nb_model.c"), "-lm"], capture_output=True, text=True)
        if build.returncode != 0 and "sanitize" in build.stderr:
            pytest.skip("this gcc has no sanitizer runtime")
        assert build.returncode == 0, build.stderr[-2000:]
        run = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS="2", ASAN_OPTIONS="detect_leaks=1"))
        assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
        assert "largest count" in run.stdout and run.stdout.count(", 0 differ") == 4
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
