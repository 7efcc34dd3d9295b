"""Shared test helpers: the reference's acceptance metric and input distribution.

compare_vectors / random_signal restate src/test_utils.rs:19-43 and tests/accuracy.rs:30-37, 84-95:
inputs re, im ~ U[0,10) from a seeded stream; pass iff mean_i |a_i - b_i| < 0.1.
(The reference's StdRng stream cannot be reproduced without the `rand` crate and is not needed: the
reference tests are differential.)
"""
import numpy as np

SEED = 0x52555354  # "RUST"


def random_signal(n, dtype, seed=SEED):
    rng = np.random.default_rng(seed + n)
    real = np.float32 if np.dtype(dtype) == np.complex64 else np.float64
    x = rng.uniform(0.0, 10.0, n).astype(real) + 1j * rng.uniform(0.0, 10.0, n).astype(real)
    return x.astype(dtype)


def zero_mean_signal(n, dtype, seed=SEED):
    rng = np.random.default_rng(seed + 7 * n + 1)
    x = rng.uniform(-1.0, 1.0, n) + 1j * rng.uniform(-1.0, 1.0, n)
    return x.astype(dtype)


def mean_abs_err(a, b):
    a = np.asarray(a).reshape(-1)
    b = np.asarray(b).reshape(-1)
    assert a.shape == b.shape
    if a.size == 0:
        return 0.0
    return float(np.mean(np.abs(a.astype(np.complex128) - b.astype(np.complex128))))


def compare_vectors(a, b):
    """tests/accuracy.rs:30-37 — THE tolerance north_star names: mean |a-b| < 0.1."""
    return mean_abs_err(a, b) < 0.1


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.complex128).reshape(-1)
    b = np.asarray(b, dtype=np.complex128).reshape(-1)
    d = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / d) if d > 0 else float(np.linalg.norm(a - b))


def numpy_fft(x, n, inverse):
    """Independent second oracle: numpy pocketfft in complex128, unnormalised both ways."""
    x = np.asarray(x, dtype=np.complex128).reshape(-1, n)
    return (np.fft.ifft(x, axis=1) * n if inverse else np.fft.fft(x, axis=1)).reshape(-1)


def check_fft_algorithm(fft, length, direction, reference=None, n=3):
    """src/test_utils.rs:70-209 restated: len/direction, then a batch of 3 through all four API
    entry points, each again with scratch pre-filled with (100,100) ("dirty scratch")."""
    assert fft.len() == length, "Algorithm reported incorrect size"
    assert fft.fft_direction() == direction, "Algorithm reported incorrect FFT direction"
    dtype = np.dtype(fft.dtype)
    x = random_signal(length * n, dtype)
    if reference is None:
        expected = numpy_fft(x, length, direction == 1) if length > 0 else x.copy()
    else:
        expected = x.copy()
        reference.process(expected)
    dirty = np.dtype(dtype).type(100 + 100j)

    buf = x.copy()
    fft.process(buf)
    assert compare_vectors(expected, buf), f"process() failed, length = {length}"

    for fill in (0, dirty):
        buf = x.copy()
        scratch = np.full(fft.get_inplace_scratch_len(), fill, dtype=dtype)
        fft.process_with_scratch(buf, scratch)
        assert compare_vectors(expected, buf), f"process_with_scratch() failed, length = {length}, fill={fill}"

    for fill in (0, dirty):
        inp = x.copy()
        out = np.zeros(n * length, dtype=dtype)
        scratch = np.full(fft.get_outofplace_scratch_len(), fill, dtype=dtype)
        fft.process_outofplace_with_scratch(inp, out, scratch)
        assert compare_vectors(expected, out), f"process_outofplace_with_scratch() failed, length = {length}"

    for fill in (0, dirty):
        inp = x.copy()
        out = np.zeros(n * length, dtype=dtype)
        scratch = np.full(fft.get_immutable_scratch_len(), fill, dtype=dtype)
        fft.process_immutable_with_scratch(inp, out, scratch)
        assert compare_vectors(expected, out), f"process_immutable_with_scratch() failed, length = {length}"
        assert np.array_equal(inp, x), "immutable input was modified"


def build_cpp_mirror_check():
    """Compiles tests/cpp/mirror_check.cpp (the C++17 host mirror's own check program) against the in-tree library."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "mirror_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(root, "tests", "cpp", "mirror_check.cpp"), "-o", exe,
                           "-L" + os.path.join(root, "rustfft_amd", "lib"), "-lmi355fft",
                           "-Wl,-rpath," + os.path.join(root, "rustfft_amd", "lib")])
    return exe


def check_host_planner_options(planner, oracle):
    """mi355fft_plan_create_ex (include/mi355fft.h): the host planner names the Recipe family, supplies its own
    compute_twiddle (src/twiddles.rs:6-23) and / or its finished Rader / Bluestein tables (raders_algorithm.rs:87-113,
    bluesteins_algorithm.rs:63-98).  Results must equal the default plan's bit for bit when the host's values are the
    library's own, and follow the host's values when they differ."""
    import pytest

    import rustfft_amd

    dtype = np.complex64
    calls = []

    def twiddle(index, fft_len):
        calls.append((index, fft_len))
        return oracle.compute_twiddle(dtype, index, fft_len, 0)  # the reference's own function, forward direction

    for n in (1024, 1200, 1 << 16, 1009, 719):
        x = random_signal(2 * n, dtype)
        want = x.copy()
        planner.plan_fft(n, 1).process(want)
        calls.clear()
        fft = planner.plan_fft_with(n, 1, twiddle_fn=twiddle)
        assert calls and all(i < 2 * l + 1 for i, l in calls), n
        got = x.copy()
        fft.process(got)
        if n in (1024, 1200, 1 << 16):
            assert np.array_equal(got, want), n  # oracle twiddles == library twiddles (both f64 angle, cos/sin, rounded to T)
        else:  # Rader / Bluestein precompute spectra FROM the twiddles: host values rounded to T move them by ~eps
            assert rel_l2(got, want) < 2e-6 and not np.array_equal(got, want), n
    # a host twiddle function that is deliberately different shows the tables really come from it
    fft = planner.plan_fft_with(4096, 0, twiddle_fn=lambda i, l: 1.0 + 0.0j)
    y = random_signal(4096, dtype)
    z = y.copy()
    fft.process(z)
    ref = y.copy()
    planner.plan_fft(4096, 0).process(ref)
    assert not np.allclose(z, ref)

    # algorithm families
    assert "bluestein" in planner.plan_fft_with(1024, 0, algorithm=rustfft_amd.ALGO_BLUESTEIN).describe()
    assert "k1<" in planner.plan_fft_with(1024, 0, algorithm=rustfft_amd.ALGO_MIXED_RADIX).describe()
    with pytest.raises(rustfft_amd.FftPanic, match="no GPU plan"):
        planner.plan_fft_with(1019, 0, algorithm=rustfft_amd.ALGO_MIXED_RADIX)  # prime, 1018 = 2 * 509: nothing direct
    # 1018 = 2 x 509, 508 = 4 x 127: until round 5 "no GPU plan"; round 6: Rader over MixedRadix over Rader ... as the reference recurses
    # (src/plan.rs:636-665), one kernel from a run-time program (the LDS stage machine)
    fr = planner.plan_fft_with(1019, 0, algorithm=rustfft_amd.ALGO_RADER)
    assert fr.describe().startswith("lsm<rader1019[mixed{rader509["), fr.describe()
    xr = random_signal(3 * 1019, dtype)
    ar, br = xr.copy(), xr.copy()
    fr.process(ar)
    oracle.plan(dtype, 1019, 0).process(br)
    assert compare_vectors(ar, br)
    x = random_signal(3 * 1024, dtype)
    a, b = x.copy(), x.copy()
    planner.plan_fft_with(1024, 0, algorithm=rustfft_amd.ALGO_BLUESTEIN).process(a)
    oracle.plan(dtype, 1024, 0).process(b)
    assert compare_vectors(a, b)

    # finished tables from the reference's own algorithm objects (here: the oracle restating them), both directions
    for d in (0, 1):
        # Bluestein: twiddles[n] and inner_fft_multiplier[M] exactly as bluesteins_algorithm.rs:63-98 builds them
        n = 719
        M = planner.bluestein_inner_len(n)
        assert M >= 2 * n - 1
        tw = np.array([oracle.compute_twiddle(dtype, (i * i) % (2 * n), 2 * n, d) for i in range(n)], dtype=dtype)
        mult = np.zeros(M, dtype=dtype)
        mult[0] = np.conj(tw[0]) / M
        for i in range(1, n):
            mult[i] = mult[M - i] = np.conj(tw[i]) / M
        inner = oracle.plan(dtype, M, d)
        inner.process(mult)
        fft = planner.plan_fft_with(n, d, algorithm=rustfft_amd.ALGO_BLUESTEIN, bluestein_twiddles=tw, bluestein_multiplier=mult)
        check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d), n=3)
        with pytest.raises(rustfft_amd.FftPanic, match="host tables do not fit"):
            planner.plan_fft_with(n, d, algorithm=rustfft_amd.ALGO_BLUESTEIN, bluestein_twiddles=tw, bluestein_multiplier=mult[: M // 2])
        # Rader: inner_fft_data[p - 1] as raders_algorithm.rs:87-113 builds it (smallest primitive root, unity / (p-1))
        p = 1009
        g = oracle.primitive_root(p)
        ginv = pow(g, p - 2, p)
        data = np.zeros(p - 1, dtype=dtype)
        t = 1
        for j in range(p - 1):
            data[j] = oracle.compute_twiddle(dtype, t, p, d) / (p - 1)
            t = t * ginv % p
        oracle.plan(dtype, p - 1, d).process(data)
        fft = planner.plan_fft_with(p, d, algorithm=rustfft_amd.ALGO_RADER, rader_inner_fft_data=data)
        assert "rader" in fft.describe()
        check_fft_algorithm(fft, p, d, reference=oracle.plan(dtype, p, d), n=3)
    with pytest.raises(rustfft_amd.FftPanic, match="needs algorithm"):
        planner.plan_fft_with(1009, 0, rader_inner_fft_data=np.zeros(1008, dtype=dtype))


def check_host_planner_recipe(planner, oracle, dtype=np.complex64, big=True):
    """mi355fft_plan_options.recipe (include/mi355fft.h): the host planner hands over its whole Recipe tree (src/plan.rs:134-188).
    The oracle's restatement of FftPlannerScalar::design_fft_for_len plays the Rust planner.  Checked: the family is the root's,
    a MixedRadix root's leaves become the pass heights (right before left, mixed_radix.rs:128-158) when tiles exist, a
    Bluesteins root's inner length is used when a kernel of that length exists, malformed trees are rejected the way the
    reference's constructors assert, and every result matches the oracle's plan."""
    import ctypes

    import pytest

    import rustfft_amd
    from rustfft_amd import Recipe, _native
    from rustfft_amd.planner import RECIPE_STATUS_FAMILY, RECIPE_STATUS_NONE, RECIPE_STATUS_SPLIT

    assert planner.plan_fft(1024, 0).recipe_status() == RECIPE_STATUS_NONE
    # 1. the scalar planner's own recipes, verbatim
    for n, family in ((1200, "k1<1200"), (1009, "rader"), (4099, "bluestein"), (97, "rader"), (10403, "k2r"), (44100, "k2g")):
        if n > 5000 and not big:
            continue
        text = oracle.recipe(n)
        for d in (0, 1):
            fft = planner.plan_fft_with(n, d, recipe=text)
            assert family in fft.describe(), (n, text, fft.describe())
            assert fft.recipe_status() >= RECIPE_STATUS_FAMILY
            check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d), n=2)
    # the reference pads 4099 to 3 * 2^12 (plan.rs:649-657); the GPU's own ladder would take 5 * 2^11
    text = oracle.recipe(4099)
    assert text.startswith("BluesteinsAlgorithm{4099,") and Recipe.parse(text).left.len == 12288
    fft = planner.plan_fft_with(4099, 0, recipe=text)
    assert fft.recipe_status() == RECIPE_STATUS_SPLIT and "12288" in fft.describe(), fft.describe()
    assert planner.bluestein_inner_len(4099) == 10240 and "10240" in planner.plan_fft(4099, 0).describe()
    # an inner length no kernel has: the GPU keeps its own, the family stays
    odd = Recipe.bluesteins(4099, Recipe.dft(8209))
    fft = planner.plan_fft_with(4099, 0, recipe=odd)
    assert fft.recipe_status() == RECIPE_STATUS_FAMILY and "10240" in fft.describe(), fft.describe()

    # 2. six-step splits chosen by the host
    def leaf(n):
        return Recipe.dft(n)

    cases = [
        (1 << 16, Recipe.mixed_radix(leaf(64), leaf(1024)), ["<1024,", "<64,"]),       # height 1024 first, then width 64
        (1 << 16, Recipe.mixed_radix(leaf(1024), leaf(64)), ["<64,", "<1024,"]),
        (1 << 18, Recipe.mixed_radix(Recipe.mixed_radix(leaf(64), leaf(64)), leaf(64)), ["<64,", "<64,", "<64,"]),
        (1517, Recipe.mixed_radix(Recipe.raders(leaf(36)), Recipe.raders(leaf(40))), ["k2rfirst<40,", "k2rlater<36,"]),  # 37 x 41: prime tiles (named by P - 1)
        (36 * 4096, Recipe.mixed_radix(leaf(36), leaf(4096), Recipe.GOOD_THOMAS), None),  # no 4096-row tile of any kind: own split
    ]
    if big:
        cases += [
            (1 << 20, Recipe.mixed_radix(leaf(512), leaf(2048)), ["<2048,", "<512,"]),
            (1 << 20, Recipe.mixed_radix(leaf(4096), leaf(256)), None),                  # no 4096-row tile: own split
            (10403, Recipe.mixed_radix(Recipe.raders(leaf(102)), Recipe.raders(leaf(100))), ["k2rfirst<100,", "k2rlater<102,"]),
        ]
    for n, tree, want in cases:
        assert tree.len == n
        d = n % 2
        fft = planner.plan_fft_with(n, d, recipe=tree)
        desc = fft.describe()
        if want is None:
            assert fft.recipe_status() == RECIPE_STATUS_FAMILY, (n, desc)
            assert desc == planner.plan_fft_with(n, d, algorithm=rustfft_amd.ALGO_MIXED_RADIX).describe()
        else:
            assert fft.recipe_status() == RECIPE_STATUS_SPLIT, (n, desc)
            kernels = desc.split(" -> ")
            assert len(kernels) == len(want) and all(w in k for w, k in zip(want, kernels)), (n, desc, want)
        check_fft_algorithm(fft, n, d, reference=oracle.plan(dtype, n, d), n=2)

    # 3. malformed trees: INVALID_ARG, never a wrong transform
    def raw(nodes, n, algorithm=0):
        arr = (_native.RecipeNode * len(nodes))()
        for i, (k, l, r, ln) in enumerate(nodes):
            arr[i].kind, arr[i].left, arr[i].right, arr[i].len = k, l, r, ln
        o = _native.PlanOptions()
        o.struct_size = ctypes.sizeof(_native.PlanOptions)
        o.algorithm = algorithm
        o.recipe = ctypes.cast(arr, ctypes.POINTER(_native.RecipeNode))
        o.recipe_nodes = len(nodes)
        h = ctypes.c_void_p()
        rc = planner._lib.mi355fft_plan_create_ex(n, 0, planner._prec, ctypes.byref(o), ctypes.byref(h))
        msg = planner._lib.mi355fft_last_error().decode()
        if rc == 0:
            planner._lib.mi355fft_plan_destroy(h)
        return rc, msg

    R = Recipe
    bad = [
        ([(R.MIXED_RADIX, 1, 2, 1024), (R.DFT, -1, -1, 32), (R.DFT, -1, -1, 16)], 1024, "left_fft.len() * right_fft.len() != len"),
        ([(R.MIXED_RADIX, 0, 1, 1024), (R.DFT, -1, -1, 32)], 1024, "after the node"),          # a child that is its own parent
        ([(R.MIXED_RADIX, 1, 5, 1024), (R.DFT, -1, -1, 32)], 1024, "after the node"),          # out of range
        ([(R.RADERS, 1, -1, 1009), (R.DFT, -1, -1, 1024)], 1009, "inner_fft.len() != len - 1"),
        ([(R.BLUESTEINS, 1, -1, 719), (R.DFT, -1, -1, 1024)], 719, "inner_fft.len() < 2 len - 1"),
        ([(R.DFT, -1, -1, 512)], 1024, "root's len"),
        ([(R.DFT, 1, -1, 1024), (R.DFT, -1, -1, 4)], 1024, "no children"),
        ([(42, -1, -1, 1024)], 1024, "unknown node kind"),
        ([(R.RADIX4, 1, -1, 1024), (R.BUTTERFLY, -1, -1, 3)], 1024, "multiple of base_fft"),
    ]
    for nodes, n, text in bad:
        rc, msg = raw(nodes, n)
        assert rc == 7 and text in msg, (nodes, rc, msg)
    rc, msg = raw([(R.RADERS, 1, -1, 1009), (R.DFT, -1, -1, 1008)], 1009, algorithm=rustfft_amd.ALGO_BLUESTEIN)
    assert rc == 7 and "another family" in msg
    rc, msg = raw([(R.RADERS, 1, -1, 1009), (R.DFT, -1, -1, 1008)], 1009, algorithm=rustfft_amd.ALGO_RADER)
    assert rc == 0
    # a recipe for a composite length under a Raders root fails like RadersAlgorithm::new's assert (raders_algorithm.rs:68)
    rc, msg = raw([(R.RADERS, 1, -1, 1025), (R.DFT, -1, -1, 1024)], 1025)
    assert rc == 6, (rc, msg)

    # 4. a binding compiled against the header BEFORE the recipe fields existed keeps working: fields beyond its struct_size
    # are not read
    o = _native.PlanOptions()
    o.struct_size = _native.PlanOptions.recipe.offset
    o.recipe_nodes = 12345  # garbage beyond the caller's struct
    h = ctypes.c_void_p()
    assert planner._lib.mi355fft_plan_create_ex(1024, 0, planner._prec, ctypes.byref(o), ctypes.byref(h)) == 0
    assert planner._lib.mi355fft_plan_recipe_status(h) == RECIPE_STATUS_NONE
    planner._lib.mi355fft_plan_destroy(h)
