"""
-m gpu: the hardware functions of the sampler's definition (PRNB-7: v_rcp_f32, v_log_f32, v_exp_f32 in both classes;
v_sqrt_f32, v_rsq_f32, v_cos_f32 in the gamma-Poisson class) as the scalar model sees them.  The model (oracle/nb_model.c) reads their values from tables written by the product's probe
kernel (prosstt_amd_hw_math); two of its lookups rest on properties of the hardware that are checked here over
EVERY argument the sampler can present:
  * v_rcp_f32(2^e * x) == v_rcp_f32(x) * 2^-e for x in [1, 2): the model keeps one table of 2^23 mantissas;
  * v_exp_f32(-x) == 1.0 for every 0 <= x < 2^-24 (denormals included): the model's table starts at 2^-24.
And the accuracy the law rests on (each function within 2e-7 of the true value), and that the probe is what the
model then returns.  The gamma-Poisson class presents arguments no table can enumerate: the model ASKS the device for
them (prosstt_amd_hw_math_at); the last tests hold the answers against binary64 over the ranges that class presents
and check that asking and tabulating agree where both exist.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rcp_scales_exactly_with_the_exponent():
    from prosstt_amd import device
    ctx = device.get_context()
    mant = ctx.hw_math("rcp", 0x3F800000, 1 << 23)                  # x in [1, 2)
    assert mant[0] == 1.0 and (mant > 0.5).all() and (mant <= 1.0).all()
    # the sampler presents 1 + theta in (1, 1e18], (1 + theta) - 1 and theta in [2^-23, 1e18], and -- the gamma-Poisson
    # class -- r in [2^-40, ...), 0.5 - |U| in [2^-34, 0.5], a/us^2 + b up to 2^66: check 2^-42 .. 2^70
    for e in range(-42, 70):
        got = ctx.hw_math("rcp", (127 + e) << 23, 1 << 23)
        want = np.ldexp(mant, -e)
        assert np.array_equal(got, want), "exponent %d: %d values differ" % (e, int((got != want).sum()))


def test_exp2_of_tiny_arguments_is_one():
    from prosstt_amd import device
    ctx = device.get_context()
    first, end = 0, 0x33800000                                      # bit patterns of [0, 2^-24)
    step = 1 << 26
    for lo in range(first, end, step):
        y = ctx.hw_math("exp2neg", lo, min(step, end - lo))
        assert (y == 1.0).all(), "v_exp_f32(-x) != 1 for some x with bits in [%#x, %#x)" % (lo, lo + step)


def test_accuracy_and_model_lookup():
    from prosstt_amd import device
    from oracle import nb_model
    assert nb_model.hw_mode()
    ctx = device.get_context()
    rng = np.random.default_rng(5)
    # log2 over (1, 25], the quotient log2(u)/(u - 1), exp2(-t2) over [0, 27.4112): relative to binary64
    u = np.concatenate([1.0 + np.exp(rng.uniform(np.log(2.0 ** -23), np.log(24.0), 400000)), [1.0 + 2.0 ** -23, 17.0, 25.0]]).astype(np.float32)
    lg = nb_model.hw_math("log2", u)
    ref = np.log2(u.astype(np.float64))
    assert np.max(np.abs(lg - ref) / np.abs(ref)) < 2e-7
    um1 = (u - np.float32(1.0)).astype(np.float32)
    quo = (lg * nb_model.hw_math("rcp", um1)).astype(np.float64)
    assert np.max(np.abs(quo / (ref / um1.astype(np.float64)) - 1)) < 4e-7
    t2 = np.concatenate([rng.uniform(0, 27.4112, 400000), np.exp(rng.uniform(np.log(1e-9), 0, 100000))]).astype(np.float32)
    ex = nb_model.hw_math("exp2neg", t2)
    assert np.max(np.abs(ex / np.exp2(-t2.astype(np.float64)) - 1)) < 2e-7
    r = nb_model.hw_math("rcp", u)
    assert np.max(np.abs(r * u.astype(np.float64) - 1)) < 1.5e-7
    # the model's lookups are the probe's values
    for op, x in (("log2", u), ("rcp", u), ("rcp", um1), ("exp2neg", t2[t2 >= 2.0 ** -24])):
        bits = x.view(np.uint32)
        pick = rng.choice(len(bits), 2000, replace=False)
        dev = np.array([ctx.hw_math(op, int(b), 1)[0] for b in bits[pick[:200]]])
        assert np.array_equal(dev, nb_model.hw_math(op, x[pick[:200]]))


def test_every_table_entry_against_binary64(capsys):
    """What protects the LAW once the hardware functions are part of the definition: every entry of the three tables the
    model was given (2^23 + 4.2e7 + 2.4e8 values -- all the sampler can present) against binary64 libm, host-side:
    rcp within 1.5e-7, log2 within 2e-7 (relative; log2(1) == 0 exactly), exp2(-x) within 2e-7; exp2(-x) non-increasing in
    x (the zero test's slack argument leans on it).  The tables' digests go into the test log so that two boxes (or two
    microcode levels) can be compared."""
    import hashlib
    from oracle import nb_model
    assert nb_model.hw_mode()
    tabs = nb_model._HW_TABLES
    worst = {}
    step = 1 << 24
    for op, (first, count) in nb_model.HW_RANGES.items():
        t = tabs[op]
        assert t.shape == (count,) and np.isfinite(t).all()
        err = 0.0
        for lo in range(0, count, step):
            hi = min(lo + step, count)
            x = (np.arange(lo, hi, dtype=np.uint32) + np.uint32(first)).view(np.float32).astype(np.float64)
            y = t[lo:hi].astype(np.float64)
            if op == "rcp":
                e = np.abs(y * x - 1.0)
            elif op == "log2":
                ref = np.log2(x)
                if lo == 0:
                    assert y[0] == 0.0                      # log2(1)
                    e = np.abs(y[1:] - ref[1:]) / ref[1:]
                else:
                    e = np.abs(y - ref) / ref
            else:
                e = np.abs(y / np.exp2(-x) - 1.0)
            err = max(err, float(e.max()))
        worst[op] = err
    assert worst["rcp"] < 1.5e-7 and worst["log2"] < 2e-7 and worst["exp2neg"] < 2e-7, worst
    ex = tabs["exp2neg"]
    for lo in range(0, ex.size - 1, step):
        seg = ex[lo:min(lo + step + 1, ex.size)]
        assert (seg[1:] <= seg[:-1]).all(), "v_exp_f32(-x) increases somewhere in table entries %d.." % lo
    assert ex[0] <= 1.0
    digest = {op: hashlib.sha256(tabs[op].tobytes()).hexdigest()[:16] for op in tabs}
    with capsys.disabled():
        print("\n[hw tables] worst relative error vs binary64: rcp %.3g, log2 %.3g, exp2neg %.3g; sha256[:16]: %s"
              % (worst["rcp"], worst["log2"], worst["exp2neg"], digest))


def test_queried_functions_against_binary64(capsys):
    """What protects the LAW of the gamma-Poisson class (PRNB-7): the hardware's log2 / sqrt / rsq / cos / exp2 over
    the arguments that class presents, against binary64 -- a million random arguments per function and range."""
    from prosstt_amd import device
    ctx = device.get_context()
    rng = np.random.default_rng(17)
    n = 1_000_000
    worst = {}
    # log2: uniforms in (0, 1], 1 + t and lam/k around 1, lam and 2 pi k up to 2^25
    x = np.concatenate([(rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -32),
                        rng.uniform(0.05, 3.0, n), np.exp(rng.uniform(np.log(1.0), np.log(3.0e7), n))]).astype(np.float32)
    y = ctx.hw_math_at("log2", x).astype(np.float64)
    ref = np.log2(x.astype(np.float64))
    worst["log2"] = float(np.max(np.abs(y - ref) / np.maximum(np.abs(ref), 2.0 ** -20)))
    assert worst["log2"] < 3e-7
    # sqrt: -2 ln u in [0, 45], lam in [10, 2^22]; rsq: 9 d, d = r - 1/3 >= 2/3
    x = np.concatenate([rng.uniform(0, 45, n), np.exp(rng.uniform(np.log(1e-6), np.log(4.2e6), n))]).astype(np.float32)
    y = ctx.hw_math_at("sqrt", x).astype(np.float64)
    worst["sqrt"] = float(np.max(np.abs(y / np.sqrt(x.astype(np.float64)) - 1)))
    assert worst["sqrt"] < 2e-7 and ctx.hw_math_at("sqrt", np.zeros(1, np.float32))[0] == 0.0
    x = np.exp(rng.uniform(np.log(6.0), np.log(1e12), n)).astype(np.float32)
    y = ctx.hw_math_at("rsq", x).astype(np.float64)
    worst["rsq"] = float(np.max(np.abs(y * np.sqrt(x.astype(np.float64)) - 1)))
    assert worst["rsq"] < 2e-7
    # cos of w / 2^32 revolutions: absolute
    w = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.float32) * np.float32(2.0 ** -32)
    y = ctx.hw_math_at("cos", w).astype(np.float64)
    worst["cos"] = float(np.max(np.abs(y - np.cos(2 * np.pi * w.astype(np.float64)))))
    assert worst["cos"] < 2e-6
    assert ctx.hw_math_at("cos", np.array([0.0, 1.0, 0.5], np.float32)).tolist() == [1.0, 1.0, -1.0]
    # exp2(-x): the boost's -log2(u)/r in [0, 2^40] and lam log2(e) < 14.5
    x = np.concatenate([rng.uniform(0, 40, n), np.exp(rng.uniform(np.log(1e-9), np.log(150.0), n))]).astype(np.float32)
    y = ctx.hw_math_at("exp2neg", x).astype(np.float64)
    ref = np.exp2(-x.astype(np.float64))
    ok = ref > 2.0 ** -120
    worst["exp2neg"] = float(np.max(np.abs(y[ok] / ref[ok] - 1)))
    assert worst["exp2neg"] < 2e-7
    assert (ctx.hw_math_at("exp2neg", np.array([200.0, 1e9, 2.0 ** 40], np.float32)) == 0.0).all()
    with capsys.disabled():
        print("\n[hw queries] worst error vs binary64 (relative; cos absolute): %s" % {k: "%.3g" % v for k, v in worst.items()})


def test_asking_equals_tabulating_and_the_model_asks():
    """prosstt_amd_hw_math_at(op, x) == prosstt_amd_hw_math(op, bits of x) for the three tabulated functions, and the
    model's gamma-Poisson class really goes through the device (query statistics of a draw)."""
    from prosstt_amd import device
    from oracle import nb_model
    ctx = device.get_context()
    for op, first in (("rcp", 0x3F800000), ("log2", 0x3F800000), ("exp2neg", 0x40400000)):
        tab = ctx.hw_math(op, first, 1 << 16)
        x = (np.arange(1 << 16, dtype=np.uint32) + np.uint32(first)).view(np.float32)
        assert np.array_equal(tab, ctx.hw_math_at(op, x))
    assert nb_model.hw_mode()
    x = nb_model.sample_iid(300.0, 0.3, 2.0, 20000, seed=3)
    rounds, values = nb_model.query_stats()
    assert rounds >= 4 and values >= 5 * 20000 and abs(x.mean() / 300.0 - 1) < 0.02
