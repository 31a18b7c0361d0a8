"""The certificates' error model of a matrix-core dot product, MEASURED.

gemm.hip.h / prescan.hip.h bound a pre-filter value's distance from the reference's by terms of the form
    | mfma_dot(a, b) - <a, b> |  <=  K * u * sum_k |a_k b_k|      (u = 2^-24, K = operand length)
i.e. "the products are exact and the accumulation errs no more than a chain of K f32 additions".  The guide states
that for the f32 MFMA; the production filters run v_mfma_f32_32x32x16_bf16 / _f16 (and 16x16x1 f32 on f32 rows), whose
internal accumulation order, rounding mode and handling of SUBNORMAL fp16 inputs are not documented.  vers_test_mfma runs
one wave of each instruction over adversarial operands, accumulating over K as the kernels do; the exact value comes from
f64 (16-bit x 16-bit products and sums of 768 of them are exact / 2^-29 below what is measured).  Asserted per pattern:
err <= K u sum|ab|; printed: the measured worst ratio err / (u sum|ab|) -- the number of worst-case roundings the hardware
actually spent -- which DESIGN.md section 1 quotes as the margin."""
import numpy as np
import pytest

from vers_amd import capi, testhooks

pytestmark = pytest.mark.gpu
U = 2.0 ** -24


def f16_bits(x):
    return np.asarray(x, dtype=np.float16).view(np.uint16)


def bf16_bits(x):
    """round-to-nearest-even f32 -> bf16 bit patterns (the split kernels' conversion)"""
    b = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((b + 0x7FFF + ((b >> 16) & 1)) >> 16).astype(np.uint16)
    return r


def bf16_to_f64(bits):
    return (bits.astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def patterns(rows, cols, K, rng, dtype):
    """(name, A [rows, K], B [K, cols]) in float64, every entry representable in `dtype` after the caller's rounding"""
    out = []
    out.append(("random normal", rng.standard_normal((rows, K)), rng.standard_normal((K, cols))))
    out.append(("unit-vector scale (|x_i| ~ 0.036), query scaled by -2", rng.standard_normal((rows, K)) / np.sqrt(K), -2.0 * rng.standard_normal((K, cols)) / np.sqrt(K)))
    out.append(("all products positive (a rounding BIAS would accumulate)", 0.5 + rng.random((rows, K)), 0.5 + rng.random((K, cols))))
    big = 60000.0 if dtype == "f16" else 3.0e38 ** 0.5
    out.append(("largest magnitudes", np.full((rows, K), big) * rng.choice([-1.0, 1.0], (rows, K)), np.full((K, cols), 1.0 / 64) * (1 + rng.random((K, cols)))))
    a = rng.standard_normal((rows, K)); b = rng.standard_normal((K, cols))
    a[:, 1::2] = -a[:, 0::2]; b[1::2, :] = b[0::2, :]            # pairs cancel exactly; one small survivor
    a[:, 0] = 2.0 ** -12; b[0, :] = 1.0; a[:, 1] = 0.0
    out.append(("cancellation (+x, -x pairs around a 2^-12 survivor)", a, b))
    if dtype == "f16":
        sub = rng.integers(1, 1024, (rows, K)) * 2.0 ** -24       # fp16 SUBNORMAL inputs (below 2^-14)
        out.append(("subnormal fp16 rows x normal queries", sub * rng.choice([-1.0, 1.0], (rows, K)), rng.standard_normal((K, cols))))
        mix = rng.standard_normal((rows, K)) / np.sqrt(K)
        mix[:, ::3] = rng.integers(1, 1024, (rows, (K + 2) // 3)) * 2.0 ** -24
        out.append(("mixed normal / subnormal row elements", mix, -2.0 * rng.standard_normal((K, cols)) / np.sqrt(K)))
        out.append(("subnormal x subnormal (products below f32's normal range)", sub, rng.integers(1, 1024, (K, cols)) * 2.0 ** -24))
    return out


@pytest.mark.parametrize("kind,name,dtype,K", [(0, "v_mfma_f32_32x32x16_f16", "f16", 768), (1, "v_mfma_f32_32x32x16_bf16", "bf16", 768),
                                               (2, "v_mfma_f32_32x32x2_f32", "f32", 768), (3, "v_mfma_f32_16x16x1_4b_f32", "f32", 768)])
def test_matrix_core_accumulation_stays_inside_the_certificates_model(kind, name, dtype, K):
    rng = np.random.default_rng(0x3FA + kind)
    rows, cols = (64, 16) if kind == 3 else (32, 32)
    worst = 0.0
    for pname, A, B in patterns(rows, cols, K, rng, dtype):
        if dtype == "f16":
            Ab, Bb = f16_bits(A), f16_bits(B)
            Ae, Be = Ab.view(np.float16).astype(np.float64), Bb.view(np.float16).astype(np.float64)
        elif dtype == "bf16":
            Ab, Bb = bf16_bits(A), bf16_bits(B)
            Ae, Be = bf16_to_f64(Ab), bf16_to_f64(Bb)
        else:
            Ab, Bb = A.astype(np.float32), B.astype(np.float32)
            Ae, Be = Ab.astype(np.float64), Bb.astype(np.float64)
        got = testhooks.mfma(kind, Ab, Bb).astype(np.float64)
        exact = Ae @ Be
        s1 = np.abs(Ae) @ np.abs(Be)
        # the result is an f32: its own final rounding is part of what is measured; a subnormal result may lose whole bits
        tiny = 2.0 ** -126
        ratio = np.abs(got - exact) / np.maximum(U * s1, tiny * 2.0 ** -23)
        r = float(ratio.max())
        print(f"{name:28s} {pname:60s} worst err / (u * sum|ab|) = {r:8.3f}   (model allows {K})")
        assert np.isfinite(got).all(), (name, pname)
        assert r <= K, (name, pname, r)
        worst = max(worst, r)
    print(f"{name}: worst ratio over all patterns {worst:.3f} of {K} allowed -> margin x{K / max(worst, 1e-9):.0f}")
