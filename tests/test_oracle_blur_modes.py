"""The oracle's three cv::GaussianBlur definitions (oracle_cvprims.cpp: legacy x86 SSE2 column pass, legacy integer, OpenCV >= 3.4.11 / 4.x
Q8.8) against an independent integer model and known answers on exact-tie pixels.  OpenCV itself is not available here: the definitions are
recalled; what these tests pin is that the oracle's literal float restatement of SymmColumnVec_32s8u equals "exact
quotient, ties to even" and where the scalar tail starts."""
import numpy as np
import pytest

from tests.blur_cases import LEGACY, CV4, blur_model, tie_image


def test_kernels(oracle):
    assert (oracle.blur_kernel(0) == LEGACY).all() and (oracle.blur_kernel(1) == LEGACY).all()
    assert (oracle.blur_kernel(2) == CV4).all() and CV4.sum() == 256 and LEGACY.sum() == 257


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("shape", [(480, 752), (61, 75), (33, 7), (9, 5), (40, 130)])
def test_blur_equals_integer_model(oracle, mode, shape):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    for img in (rng.integers(0, 256, shape, dtype=np.uint8), rng.integers(120, 136, shape, dtype=np.uint8),
                np.full(shape, 255, np.uint8), np.zeros(shape, np.uint8)):
        with oracle.cv_mode(mode):
            got = oracle.blur(img)
        want, _ = blur_model(img, mode)
        assert (got == want).all()


def test_constant_images(oracle):
    for c in (0, 1, 77, 128, 200, 255):
        img = np.full((32, 40), c, np.uint8)
        with oracle.cv_mode(1):
            assert (oracle.blur(img) == min((c * 257 * 257 + 32768) >> 16, 255)).all()    # legacy kernel sums to 257: 128 -> 129
        with oracle.cv_mode(2):
            assert (oracle.blur(img) == c).all()                                           # the Q8.8 kernel sums to 256: identity


@pytest.mark.parametrize("w", [70, 71, 72, 73, 143])
def test_tie_known_answers(oracle, w):
    """Columns 7j+3 of the tie image hold sum = 128.5 * 65536 exactly: 129 in integer mode everywhere; 128 in SSE2 mode inside the vector
    body [0, w & ~3) and 129 on the scalar tail; the Q8.8 kernel has no tie there."""
    img = tie_image(3, w, 24)
    _, s = blur_model(img, 1)
    cols = np.arange(3, w - 3, 7)
    assert ((s[:, cols] & 0xFFFF) == 0x8000).all() and ((s[:, cols] >> 16) == 128).all()
    with oracle.cv_mode(1):
        b_int = oracle.blur(img)
    with oracle.cv_mode(0):
        b_sse = oracle.blur(img)
    assert (b_int[:, cols] == 129).all()
    body = cols[cols < (w & ~3)]
    tail = cols[cols >= (w & ~3)]
    assert (b_sse[:, body] == 128).all()
    assert (b_sse[:, tail] == 129).all()
    other = np.setdiff1d(np.arange(w), cols)
    assert (b_sse[:, other] == b_int[:, other]).all()
    with oracle.cv_mode(2):
        assert (oracle.blur(img) == blur_model(img, 2)[0]).all()


@pytest.mark.parametrize("w", [69, 70, 71, 72])
def test_tail_tie(oracle, w):
    """An exact tie on the last column: inside the SSE2 body only when w % 4 == 0."""
    img = tie_image(11, w, 16, tail_tie=True)
    _, s = blur_model(img, 1)
    assert ((s[:, w - 1] & 0xFFFF) == 0x8000).all() and ((s[:, w - 1] >> 16) == 128).all()
    with oracle.cv_mode(0):
        b = oracle.blur(img)
    assert (b[:, w - 1] == (128 if w % 4 == 0 else 129)).all()
    with oracle.cv_mode(1):
        assert (oracle.blur(img)[:, w - 1] == 129).all()


def test_descriptors_depend_on_mode(oracle):
    """On the tie image the three modes give three different descriptor sets for the same keypoints (the GPU tier then demands equality
    with the oracle mode by mode, tests/test_gpu_blur_modes.py)."""
    img = tie_image(5, 160, 120)
    ex = oracle.Extractor(500, 1.2, 1, 20, 7)
    ds = []
    for mode in (0, 1, 2):
        with oracle.cv_mode(mode):
            b = oracle.blur(img)
        ds.append(np.stack([ex.descriptor(b, x, 60.0, ang) for x in range(30, 130, 9) for ang in (0.0, 33.0, 90.0, 217.5)]))
    assert (ds[0] != ds[1]).any() and (ds[1] != ds[2]).any() and (ds[0] != ds[2]).any()
