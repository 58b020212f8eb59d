"""Live comparisons against the imported reference -- only where /root/reference exists (the build container; skipped on the GPU
box, which has no reference).  Nothing here is a GPU test; fixtures are not needed because both sides run in the same process."""
import glob
import json
import os
import sys
import tempfile

import numpy as np
import pytest

REF = os.environ.get("UHC_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "uhc")), reason="the reference only exists in the build container")

_PATHS = ("cfg_dict", "base_dir", "cfg_dir", "model_dir", "result_dir", "log_dir", "tb_dir", "output", "output_dir", "data_dir", "mujoco_model_file",
          "vis_model_file", "main_result_dir")


def _same(v, w):
    if isinstance(v, np.ndarray) or isinstance(w, np.ndarray):
        return np.array_equal(np.asarray(v, dtype=float), np.asarray(w, dtype=float), equal_nan=True)
    if isinstance(v, float) and np.isinf(v):
        return bool(np.isinf(w))
    return v == w


def test_config_equals_reference_on_every_shipped_config():
    """Every yml under the reference's config/ that the reference's own Config can load (85 of 115: the rest name model files it does
    not ship) gives the same attributes through uhc_amd's Config -- a yml written for the reference loads unchanged."""
    import yaml
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import ref_import
    import uhc  # noqa: F401  -- this build's alias package on purpose: the comparison must survive it being imported first
    with ref_import.reference_modules():
        from uhc.utils.config_utils.copycat_config import Config as RefConfig
        ref_import.assert_is_reference(RefConfig)
    from uhc_amd.utils.config_utils.copycat_config import Config as MyConfig
    assert RefConfig is not MyConfig
    assert os.path.realpath(RefConfig.__init__.__code__.co_filename).startswith(os.path.realpath(REF) + os.sep)
    assert os.sep + "uhc_amd" + os.sep in os.path.realpath(MyConfig.__init__.__code__.co_filename)
    assert sys.modules["uhc"] is uhc  # the alias is back for whoever runs after this test
    files = sorted(glob.glob(os.path.join(REF, "config", "**", "*.yml"), recursive=True))
    assert len(files) > 100
    compared = 0
    for f in files:
        cid = os.path.splitext(os.path.basename(f))[0]
        cd = yaml.safe_load(open(f))
        base = tempfile.mkdtemp()
        os.symlink(os.path.join(REF, "assets"), os.path.join(base, "assets"))
        try:
            rc = RefConfig(cfg_id=cid, base_dir=base, cfg_dict=json.loads(json.dumps(cd)))
        except OSError:
            continue  # the config names a model file the reference does not ship
        mc = MyConfig(cfg_id=cid, base_dir=tempfile.mkdtemp(), cfg_dict=json.loads(json.dumps(cd)))
        for k, v in vars(rc).items():
            if k in _PATHS:
                continue
            assert hasattr(mc, k), (f, k)
            assert _same(v, getattr(mc, k)), (f, k, v, getattr(mc, k))
        compared += 1
    assert compared >= 80
