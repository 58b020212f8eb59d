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
    ref_import.install()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from uhc.utils.config_utils.copycat_config import Config as RefConfig
    from uhc_amd.utils.config_utils.copycat_config import Config as MyConfig
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
