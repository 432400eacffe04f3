"""The CifHr tile kernel (csrc/cifhr.hip, apply_cell_tile) computes the reference's ``-0.5 * d2 / sigma2`` (cif_hr.cpp:81, a double
division rounded to float = the correctly rounded float quotient) in three instructions instead of the compiler's eleven:

    r   = RN(1 / sigma2)                 once per cell
    q0  = a * r ;  rem = fma(-sigma2, q0, a) ;  q = fma(rem, r, q0)

Markstein's theorem makes q the correctly rounded quotient when r is correctly rounded, q0 is within one ulp and the
significand of sigma2 is not all ones (such a cell takes the compiler's division).  This test checks the identity -- and that
q0 never is more than one ulp off -- on operand pairs of the kernel's domain: sigma2 = sigma * sigma >= 1, a = -0.5 * (dx^2 + dy^2)
with pixel-minus-centre offsets, d2 <= sigma2.  float32 fma is modelled in float64: the product of two float32 is exact there,
and the one rounding of the sum to float64 in front of the rounding to float32 can only matter within 2^-29 of a tie.
The kernel itself is checked by the bit-exact map tests (tests/test_gpu_parity.py) and the randomised sweeps."""
import numpy as np


def f32(x):
    return np.asarray(x, dtype=np.float32)


def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def check(sigma, rng, n):
    b = f32(sigma * sigma)
    bx, by = f32(rng.uniform(0, 641, n)), f32(rng.uniform(0, 641, n))
    reach = np.ceil(sigma.astype(np.float64)) + 1                  # pixels of the cell's box
    dx = f32(f32(np.floor(bx) + np.floor(rng.uniform(-reach, reach + 1))) - bx)
    dy = f32(f32(np.floor(by) + np.floor(rng.uniform(-reach, reach + 1))) - by)
    d2 = f32(f32(dx * dx) + f32(dy * dy))
    all_ones = (b.view(np.uint32) & 0x7FFFFF) == 0x7FFFFF
    keep = (d2 <= b) & (d2 > 0) & ~all_ones
    a = f32(-0.5) * d2
    r = f32(np.float64(1.0) / b.astype(np.float64))
    q0 = f32(a * r)
    q = fma32(fma32(-b, q0, a), r, q0)
    want = (a.astype(np.float64) / b.astype(np.float64)).astype(np.float32)
    ulps = np.abs(q0.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
    assert int(keep.sum()) > n // 10
    assert not (keep & (q != want)).any()
    assert not (keep & (ulps > 1)).any()
    return int(keep.sum())


def test_three_instruction_quotient_is_correctly_rounded():
    rng = np.random.default_rng(7)
    n = 2_000_000
    total = 0
    total += check(f32(rng.uniform(1.0, 60.0, n)), rng, n)                      # the sigmas of 641-px fields and beyond
    total += check(f32(rng.uniform(1.0, 3.0, n)), rng, n)                       # small boxes
    total += check(f32(np.exp(rng.uniform(0.0, np.log(2000.0), n))), rng, n)    # log-uniform, far beyond any field
    # significands next to the exception (all ones) and next to powers of two
    edge = np.concatenate([(np.uint32(0x3F800000) + np.arange(0, 64, dtype=np.uint32)),
                           (np.uint32(0x40FFFFFF) - np.arange(0, 64, dtype=np.uint32))]).view(np.float32)
    total += check(f32(np.sqrt(rng.choice(edge, n).astype(np.float64) * rng.choice([1.0, 4.0, 16.0, 64.0], n))), rng, n)
    assert total > 1_500_000
