"""Run in the build container only (needs /root/reference): compiles the reference's MJCF + STL
assets into this build's flat model format and converts the shipped sample clip to npz.
Outputs are numeric data; no reference source is copied."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from uhc_amd.model.mjcf import compile_mjcf_file  # noqa: E402

REF = os.environ.get("UHC_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "uhc_amd", "assets")


def main():
    os.makedirs(OUT, exist_ok=True)
    m = compile_mjcf_file(os.path.join(REF, "assets/mujoco_models/humanoid_smpl_neutral_mesh.xml"))
    m.save(os.path.join(OUT, "humanoid_smpl_neutral_mesh.npz"))
    import joblib
    d = joblib.load(os.path.join(REF, "sample_data/standing_neutral.pkl"))
    np.savez_compressed(os.path.join(OUT, "standing_neutral.npz"), **{k: np.asarray(v, dtype=np.float64) for k, v in d.items()})
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
