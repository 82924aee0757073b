"""csrc/exact_arith.hpp - the short correctly-rounded quotient / reciprocal / square-root sequences the map kernels' fragment stages use -
against the compiler's IEEE sequences ON THE DEVICE, operand by operand over whole ranges (dms_exact_arith_selftest): the one-argument
functions exhaustively over their stated domains, the quotient for every finite numerator and a set of divisors that holds every camera
constant of the test suite and the bench (fx, fy, 2 x depth cut-off) beside awkward ones (all-ones and near-one significands, tiny, huge)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dms():
    from densemonoslam_amd import capi

    assert capi.device_count() >= 1, "no MI355X visible"
    return capi


def _run(dms, what, d=1.0):
    bad, first = C.c_ulonglong(0), C.c_uint(0)
    rc = dms.lib.dms_exact_arith_selftest(what, float(d), C.byref(bad), C.byref(first))
    assert rc == 0, dms.lib.dms_last_error()
    return bad.value, first.value


def test_square_root_of_every_normal_float(dms):
    assert _run(dms, 0) == (0, 0)


def test_reciprocal_over_one_to_four(dms):
    assert _run(dms, 1) == (0, 0)


DIVISORS = [528.0, 535.4, 517.3, 516.5, 264.0, 264.2, 132.0, 718.856, 481.2, 480.0, 6.0, 8.0, 80.0, 20.0, 40.0, 3.0, 1.0, 2.0,
            float(np.float32(2.0) - np.float32(2.0 ** -23)), float(np.float32(1.0) + np.float32(2.0 ** -23)), 1.0e-3, 12345.678, 1.0 / 3.0, -535.4,
            3.0e20, 7.0e-21]


@pytest.mark.parametrize("d", DIVISORS)
def test_quotient_by_a_constant_for_every_finite_numerator(dms, d):
    assert _run(dms, 2, np.float32(d)) == (0, 0)


def test_random_divisors(dms):
    rng = np.random.default_rng(20260930)
    for d in np.exp(rng.uniform(np.log(1e-3), np.log(1e4), 24)).astype(np.float32):
        assert _run(dms, 2, d) == (0, 0), float(d)
