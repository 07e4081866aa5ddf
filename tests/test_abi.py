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
