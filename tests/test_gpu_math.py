"""The kernels' 1/x and sqrt(x) are short instruction sequences (csrc/pt_math.h: v_rcp_f32 / v_rsq_f32 + one fused correction)
for operands within [2^-100, 2^100] and the compiler's IEEE expansions elsewhere. The arithmetic contract says they ARE the
correctly rounded IEEE results — what the CPU oracle computes with '/' and sqrtf. That is checked here on every float there is:
the library runs both forms over all 2^32 bit patterns on the device and counts the inputs whose results differ in any bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("which,name", [(0, "1/x"), (1, "sqrt(x)"), (2, "1/x of the triangle test (|x| >= 1e-6)")])
def test_short_forms_equal_the_ieee_expansions_on_all_floats(gpu_ctx, which, name):
    n_diff, first = gpu_ctx.debug_exact_math(which)
    assert n_diff == 0, f"{name}: {n_diff} of 2^32 inputs differ from the IEEE result, the first is bit pattern {first:#010x}"


def test_short_forms_against_numpy_on_a_sample(gpu_ctx):
    """... and against the host's IEEE arithmetic (numpy float32 division / sqrt are correctly rounded), on values that exercise
    both branches: the short form's range, its edges, denormals, huge values, zeros, infinities."""
    rng = np.random.default_rng(5)
    x = np.concatenate([
        rng.uniform(-4, 4, 200000).astype(np.float32),
        np.exp2(rng.uniform(-149, 128, 200000)).astype(np.float32) * rng.choice([-1, 1], 200000).astype(np.float32),
        np.float32([0.0, -0.0, np.inf, -np.inf, 2.0 ** -100, 2.0 ** 100, np.nextafter(np.float32(2.0 ** -100), np.float32(0)),
                    np.nextafter(np.float32(2.0 ** 100), np.float32(np.inf)), 1e-45, 3.4e38, 1.0, 3.0]),
    ])
    with np.errstate(all="ignore"):
        want_r = (np.float32(1.0) / x).astype(np.float32)
        got_r = gpu_ctx.debug_math(12, x)
        assert np.array_equal(got_r.view(np.uint32), want_r.view(np.uint32))
        pos = np.abs(x)
        want_s = np.sqrt(pos).astype(np.float32)
        got_s = gpu_ctx.debug_math(1, pos)
        assert np.array_equal(got_s.view(np.uint32), want_s.view(np.uint32))
