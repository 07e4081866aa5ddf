"""The committed tests/golden/*_ref.npz files are outputs of the reference's own code (oracle/_ref/*.so).  Where those libraries exist (the build
container, with the reference checkout) every generator is run again into a temporary file and must reproduce the committed bytes: a golden file
cannot silently drift away from the reference it claims to record.  CPU tier; skipped where oracle/_ref was never built."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import oracle_py as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GENERATORS = [("make_golden_triangulation", "triangulation_ref.npz"), ("make_golden_matcher_ref", "matcher_ref.npz"), ("make_golden_extract_ref", "extract_ref.npz"),
              ("make_golden_frame_ref", "frame_ref.npz"), ("make_golden_stereo_bow_ref", "stereo_bow_ref.npz"), ("make_golden_direct_ref", "direct_ref.npz"),
              ("make_golden_align_ref", "align_ref.npz")]
HAVE = all(f() is not None for f in (O.ref_matcher_lib, O.ref_extractor_lib, O.ref_frame_lib, O.ref_mappoint_lib, O.ref_dbow2_lib))

pytestmark = pytest.mark.skipif(not HAVE, reason="oracle/_ref libraries not built (reference checkout absent)")


@pytest.mark.parametrize("tool,golden", GENERATORS)
def test_generator_reproduces_the_committed_file(tool, golden, tmp_path, capsys):
    spec = importlib.util.spec_from_file_location(tool, os.path.join(ROOT, "tools", tool + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = str(tmp_path / golden)
    mod.main(out)
    capsys.readouterr()
    new, old = np.load(out), np.load(os.path.join(ROOT, "tests", "golden", golden))
    assert sorted(new.files) == sorted(old.files)
    for k in new.files:
        a, b = new[k], old[k]
        assert a.dtype == b.dtype and a.shape == b.shape, k
        assert np.array_equal(a.view(np.uint8) if a.dtype.kind == "f" else a, b.view(np.uint8) if b.dtype.kind == "f" else b), k
