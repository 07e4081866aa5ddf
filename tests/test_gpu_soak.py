"""Long-run hygiene of the library: device memory must not grow while the same calls repeat (contexts grow their scratch buffers once and
keep them), and creating / destroying contexts returns everything."""
import numpy as np
import pytest

from orb_ygz_slam_amd.synth import synth_frame
from orb_ygz_slam_amd.scene import two_view_scene

pytestmark = pytest.mark.gpu


def _free_bytes(ex):
    return ex.device_mem_info()[0]      # (the library's own HIP runtime: a second copy loaded through ctypes may not see the device)


def test_repeated_calls_do_not_grow_device_memory():
    from orb_ygz_slam_amd import Extractor, make_camera, EUROC
    w, h = 752, 480
    cam = make_camera(w, h)
    ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=4)
    imgs = np.stack([synth_frame(60 + i, w, h) for i in range(4)])
    A, B, _, bp = two_view_scene(9, w, h, EUROC)
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    inv = ex.tables()["inv_scale"]

    def one_round(i):
        ex.extract_batch_host(imgs)
        ex.match_batch_prev(cam, 15.0, True, True, True)
        ex.align_batch_prev(cam, 7, 1, 10)
        ex.batch_fetch(i % 4)
        k, d = ex.extract(A if i % 2 else B)
        pa, pb = ex.compute_pyramid(A), ex.compute_pyramid(B)
        ex.sia_run(cam, k, bp(k["x"], k["y"]), ident, pa, ident, pb, inv, 7, 1)
        ex.features_in_area(cam, k, np.array([[300, 200, 40]], np.float32))
        ex.descriptor_distance(d[:100], d[100:200])

    for i in range(6):                       # every buffer reaches its working size
        one_round(i)
    before = _free_bytes(ex)
    for i in range(150):
        one_round(i)
    after = _free_bytes(ex)
    assert before - after < (8 << 20), (before, after)      # allocator slack only: no growth with the call count


def test_contexts_give_their_memory_back():
    from orb_ygz_slam_amd import Extractor
    w, h = 752, 480
    img = synth_frame(61, w, h)
    for _ in range(3):                       # the first contexts load code objects and warm the allocator
        ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=8)
        ex.extract(img)
        ex.close()
    probe = Extractor(100, 1.2, 4, 20, 7, max_width=64, max_height=64, max_batch=1)   # a small context that only reports
    before = _free_bytes(probe)
    for _ in range(40):
        ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=8)
        ex.extract(img)
        ex.close()
    after = _free_bytes(probe)
    assert before - after < (8 << 20), (before, after)


def test_awkward_widths_do_not_fall_off_the_copy_fast_path(oracle):
    """Pitched host<->device copies whose width is not a multiple of 4 are executed row by row by the copy engine (3 ms per 1241x376 frame,
    13 ms per pyramid read-back); the library re-pitches on the device instead.  Latency of such sizes must stay in the family of the
    752x480 case, and the results must still be the oracle's."""
    import time
    from orb_ygz_slam_amd import Extractor

    def med(f, n=30):
        for _ in range(5):
            f()
        t = []
        for _ in range(n):
            t0 = time.perf_counter()
            f()
            t.append(time.perf_counter() - t0)
        return float(np.median(t))

    lat = {}
    for (w, h) in ((752, 480), (1241, 376), (641, 479), (1242, 375)):
        ex = Extractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
        img = synth_frame(5, w, h)
        lat[(w, h)] = (med(lambda: ex.extract(img)), med(lambda: ex.compute_pyramid(img)))
        k, d = ex.extract(img)
        ok, od = oracle.Extractor(1000, 1.2, 8, 20, 7).extract(img)
        assert np.array_equal(k["x"], ok["x"]) and np.array_equal(k["y"], ok["y"]) and np.array_equal(d, od), (w, h)
        strided = np.zeros((h, w + 13), np.uint8)                      # a host image with a row pitch that is not its width
        strided[:, :w] = img
        k2, d2 = ex.extract(strided[:, :w])
        assert np.array_equal(k2, k) and np.array_equal(d2, d), (w, h)
    base_e, base_p = lat[(752, 480)]
    for key, (e, p) in lat.items():
        assert e < 3 * base_e + 2e-4 and p < 3 * base_p + 2e-4, (key, lat)
    print({k: (round(1e6 * a), round(1e6 * b)) for k, (a, b) in lat.items()})
