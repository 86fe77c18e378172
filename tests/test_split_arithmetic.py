"""The arithmetic contract of the split-fp16 convolution kernels (happypose_amd/csrc/conv_split.hip,
conv_igemm_split.hip), restated in NumPy so that it is checked without a GPU:

    x = x_hi + x_lo,  x_hi = fp16(x),  x_lo = fp16(x - x_hi);  w likewise after a power-of-two scaling per
    output channel;  x.w ~= sum(x_hi w_hi) + sum(x_hi w_lo) + sum(x_lo w_hi)   accumulated in fp32.

The GPU parity tests (tests/test_gpu_kernels.py::test_conv3x3_kernel_families[split-*]) hold the kernels to
2e-5 of max|ref| against fp64 -- the bound of the exact-fp32 direct kernels; this file shows that the bound
follows from the scheme itself (and what breaks it: no low halves, no weight scaling with tiny weights)."""

import numpy as np


def _split(v):
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def _scale_pow2(w):
    """per-row power of two that puts max|w| into [2^13, 2^14) (what split_weights_kernel does)."""
    mx = np.abs(w).max(axis=1)
    e = np.frexp(mx)[1]  # mx = m 2^e, m in [0.5, 1)
    s = np.where(mx > 0, 14 - e, 0)
    return np.ldexp(np.ones_like(mx), s).astype(np.float32)


def _split_matmul(x, w, scale_weights=True, low_halves=True):
    sc = _scale_pow2(w) if scale_weights else np.ones(w.shape[0], np.float32)
    xh, xl = _split(x)
    wh, wl = _split(w * sc[:, None])
    y = xh @ wh.T  # fp32 accumulation of exact fp16 x fp16 products, like v_mfma_f32_32x32x16_f16
    if low_halves:
        y = y + xh @ wl.T + xl @ wh.T
    return y / sc[None, :]


def _rel_err(y, x, w):
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    return np.abs(y - ref).max() / np.abs(ref).max()


def test_split_product_meets_fp32_bound():
    rs = np.random.RandomState(0)
    for k, wscale in ((576, 0.05), (2304, 0.02), (4608, 1e-4), (160, 3.0)):
        x = np.maximum(rs.normal(size=(256, k)), 0).astype(np.float32)  # post-ReLU activations
        w = (rs.normal(size=(64, k)) * wscale).astype(np.float32)
        assert _rel_err(_split_matmul(x, w), x, w) < 2e-6  # an order of magnitude inside the kernels' stated 2e-5
        fp32 = _rel_err((x @ w.T).astype(np.float32), x, w)
        assert _rel_err(_split_matmul(x, w), x, w) < 8 * max(fp32, 1e-7)  # fp32-level, not fp16-level


def test_split_needs_low_halves_and_weight_scaling():
    rs = np.random.RandomState(1)
    x = np.maximum(rs.normal(size=(128, 1152)), 0).astype(np.float32)
    w = (rs.normal(size=(32, 1152)) * 0.02).astype(np.float32)
    assert _rel_err(_split_matmul(x, w, low_halves=False), x, w) > 2e-5  # plain fp16 operands miss the bound
    tiny = (w * 1e-6).astype(np.float32)  # |w| ~ 2e-8: hi halves are fp16 subnormals without the scaling
    assert _rel_err(_split_matmul(x, tiny, scale_weights=False), x, tiny) > 1e-3
    assert _rel_err(_split_matmul(x, tiny), x, tiny) < 2e-6


def test_small_activations_survive_as_fp16_subnormals():
    # |x| << 2^-3: the low half falls below the fp16 normal range; the kernels rely on the f16 MFMA honouring
    # subnormals (tools/probes/mfma_f16_denorm.hip).  The absolute error stays below 2^-25 per element.
    x = np.float32([1e-3, 3e-5, 7e-7, 0.1234567, 5.0e-8])
    hi, lo = _split(x)
    assert np.abs(x - (hi + lo)).max() <= 2.0 ** -25
