"""CPU tier: the C-ABI shared library loads without a GPU and exports every symbol include/ygzf.h declares; creating a
context without a HIP device fails loudly (there is no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from tests.conftest import ROOT, have_gpu


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "ygzf.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b(ygzf_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("ygzf_create", "ygzf_destroy", "ygzf_extract", "ygzf_extract_batch_device", "ygzf_compute_pyramid",
                 "ygzf_search_by_projection_last", "ygzf_match_batch_prev", "ygzf_descriptor_distance", "ygzf_sia_run"):
        assert must in names, must


def test_library_exports_every_declared_symbol():
    from orb_ygz_slam_amd import load_library
    L = load_library()
    missing = [n for n in declared_functions() if not hasattr(L, n)]
    assert not missing, "declared in include/ygzf.h but not exported: %s" % missing


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "ygzf.h"\nint main(void){ ygzf_extractor_cfg c = {1000, 1.2f, 8, 20, 7}; (void)c; return sizeof(ygzf_kp) == 28 ? 0 : 1; }\n')
    exe = tmp_path / "t"
    import subprocess
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0


def test_no_cpu_fallback():
    if have_gpu():
        pytest.skip("GPU present")
    from orb_ygz_slam_amd import Extractor, YgzfError
    with pytest.raises(YgzfError) as e:
        Extractor()
    assert "no HIP device" in str(e.value) or "-2" in str(e.value)


def test_product_does_not_reference_the_oracle():
    """The product tree must not include, import or link anything under oracle/."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "orb_ygz_slam_amd")):
        for f in files:
            if f.endswith((".hip", ".h", ".cpp", ".cc", ".py")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r'^\s*#include\s*[<"][^>"\n]*oracle|^\s*from\s+oracle\b|^\s*import\s+oracle\b|libygz_oracle', txt, flags=re.M):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_device_index_is_never_silently_remapped():
    """ygzf_create(device = k) on a host without that device: YGZF_ERR_NO_DEVICE for every index, no fall-back to another device or to the CPU."""
    if have_gpu():
        pytest.skip("GPU present (the live-device form of this check is tests/test_gpu_errors.py)")
    from orb_ygz_slam_amd.capi import ExtractorCfg, load_library
    L = load_library()
    for dev in (0, 1, 7, -1):
        h = C.c_void_p()
        assert L.ygzf_create(dev, C.byref(ExtractorCfg(1000, 1.2, 8, 20, 7, 0)), 640, 480, 1, C.byref(h)) == -2 and not h.value


def test_scale_tables_without_a_device(oracle):
    """ORBextractor's constructor tables come from host arithmetic alone (ygzf_scale_tables_host): Frame's constructors read them before any
    image -- and any device -- is involved.  Equal to the oracle's (= the reference constructor's, tests/test_ref_extractor.py)."""
    import numpy as np
    from orb_ygz_slam_amd.capi import ExtractorCfg, load_library
    L = load_library()
    for nf, sf, nl in ((1000, 1.2, 8), (2000, 1.2, 8), (8000, 1.2, 12), (500, 2.0, 4), (1234, 1.1, 16)):
        sc, inv, s2, is2 = (np.zeros(nl, np.float32) for _ in range(4))
        nfeat = np.zeros(nl, np.int32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        assert L.ygzf_scale_tables_host(C.byref(ExtractorCfg(nf, sf, nl, 20, 7, 0)), vp(sc), vp(inv), vp(s2), vp(is2), vp(nfeat)) == 0
        t = oracle.Extractor(nf, sf, nl, 20, 7).tables()
        assert (sc == t["scale"]).all() and (inv == t["inv_scale"]).all() and (s2 == t["sigma2"]).all() and (is2 == t["inv_sigma2"]).all()
        assert (nfeat == t["nfeat"]).all()


def test_host_layout_helpers_without_a_device():
    """ygzf_host_row_pitch is host arithmetic (the device's level-0 row pitch: w rounded up to 64); ygzf_bind_host_thread_to_device names a device
    that is not there as such and leaves the caller's affinity alone."""
    from orb_ygz_slam_amd.capi import bind_host_thread_to_device, host_row_pitch
    assert [host_row_pitch(w) for w in (752, 640, 1920, 3840, 1, 65)] == [768, 640, 1920, 3840, 64, 128]
    assert host_row_pitch(0) < 0
    if have_gpu():
        pytest.skip("GPU present (bench.py binds its rank there)")
    before = os.sched_getaffinity(0)
    assert bind_host_thread_to_device(0) == -2 and bind_host_thread_to_device(5) == -2
    assert os.sched_getaffinity(0) == before
