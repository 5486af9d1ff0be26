"""The parity arm's short square root and reciprocal square root (tn_math.h sqrt_candidate / rsqrt_candidate) are PROVEN, not
sampled: the device compares them with the compiler's correctly rounded sqrtf(x) and 1.0f/sqrtf(x) on every one of the 2^32 fp32
bit patterns (tinsel_hip_selftest_arith).  The same harness shows that it has teeth: the un-scaled root and the straight-line
reciprocal differ from IEEE on denormal / huge operands, and the reciprocal root started from v_rsq_f32's own value is wrong on
exactly 255 operands of 2^32 (a Newton step from below that lands on a tie) -- which is why the library is not built with them."""
import pytest

import tinsel_amd

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("op,name", [(0, "1.0f/x"), (1, "sqrtf(x)"), (2, "1.0f/sqrtf(x)")])
def test_sequences_the_library_is_built_with_equal_ieee_on_all_inputs(op, name):
    counts, first = tinsel_amd.selftest_arith(op)
    print("%s as built: %d mismatches over 2^32 operands" % (name, counts[0]))
    assert counts[0] == 0 and first == 0xffffffff and sum(counts[4:]) == 0


@pytest.mark.parametrize("op,variant", [(1, 21), (1, 11), (0, 11), (2, 0), (2, 2), (2, 3)])
def test_proven_variants(op, variant):
    counts, first = tinsel_amd.selftest_arith(op, variant)
    assert counts[0] == 0 and first == 0xffffffff


def test_harness_finds_the_sequences_that_are_not_exact():
    # v_rsq + one Newton step without the 2^32 scaling: the residual x - s*s underflows below 2^-96; denormal operands are flushed
    counts, first = tinsel_amd.selftest_arith(1, 1)
    assert counts[1] == 2**24 - 2 and counts[3] > 0 and first == 1
    assert sum(counts[4 + 32:]) == 0                       # exponent fields 32 and up: exact
    # v_rcp + two Newton steps + v_div_fixup: wrong on denormal operands and where the quotient is denormal
    counts, first = tinsel_amd.selftest_arith(0, 1)
    assert counts[0] > 0 and counts[3] == 0 and counts[1] + counts[2] == counts[0]
    # 1/sqrt with v_rsq_f32's value as the first guess of the root's reciprocal: two operands per even exponent, where the root's
    # mantissa is all ones, and the largest denormal
    counts, first = tinsel_amd.selftest_arith(2, 1)
    assert counts[0] == 255 and first == 0x007fffff
