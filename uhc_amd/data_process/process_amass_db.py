"""AMASS database -> the clip dictionaries DatasetAMASSSingle reads (reference: uhc/data_process/process_amass_db.py:221-368).

Input: the joblib dict `amass_db_smplx.pt` {name: {poses (T, >=72), trans (T, 3), betas, gender, mocap_framerate}} and the occlusion
annotations `sample_data/amass_copycat_occlusion_v2.pkl` {"0-" + name: {issue, idxes}}.  Per sequence (process_qpos_list, :221-291):
  * subsample to 30 fps by integer stride int(mocap_framerate / 30);
  * annotated sequences: issue "sitting" / "airborne" with idxes -> truncated at idxes[0] (dropped if that leaves < 10 frames),
    any other issue -> dropped;  sequences shorter than 10 frames -> dropped;
  * height fix (fix_height_smpl_vanilla, :199-219): shift trans z so that the lowest SMPL vertex of frame 0 touches z = 0:
    `fix_height=uhc_amd.smpllib.smpl_robot.make_fix_height(body_provider)` (SMPLBody over the licensed SMPL files, which this image
    lacks, or any (betas, gender) -> (vertices, joints, skin weights) provider; linear blend skinning without the pose blend shapes);
    without the hook trans is kept and the fact is recorded under key "height_fixed": False;
  * pose_6d = first two columns of every joint's rotation matrix, float32 like the reference's convert_aa_to_orth6d
    (uhc/utils/transform_utils.py:91-100, 76-78).
Split (:294-368): by the data-set prefix of the sequence name; as in the reference the table spells the validation split
"vald" while the assignment tests for "valid", so validation sequences land in the training dictionary."""
import argparse

import numpy as np

AMASS_SPLITS = {
    "vald": ["HumanEva", "MPI_HDM05", "SFU", "MPI_mosh"],
    "test": ["Transitions_mocap", "SSM_synced"],
    "train": ["CMU", "MPI_Limits", "TotalCapture", "Eyes_Japan_Dataset", "KIT", "BML", "EKUT", "TCD_handMocap", "BMLhandball", "DanceDB",
              "ACCAD", "BMLmovi", "BioMotionLab", "Eyes", "DFaust"],
}
AMASS_SPLIT_DICT = {d: k for k, v in AMASS_SPLITS.items() for d in v}


def convert_aa_to_orth6d(pose_aa):
    """(T, 72) axis-angle -> (T, 24, 6) float32: per joint [R00 R10 R20 R01 R11 R21] (the first two columns), computed in float32 with
    the small-angle branch of the reference's angle_axis_to_rotation_matrix (theta^2 <= 1e-6 -> first-order matrix)."""
    aa = np.asarray(pose_aa, dtype=np.float32).reshape(-1, 3)
    theta2 = (aa * aa).sum(1, dtype=np.float32)
    theta = np.sqrt(theta2)
    w = aa / (theta + np.float32(1e-6))[:, None]
    c, s = np.cos(theta), np.sin(theta)
    one = np.float32(1.0)
    wx, wy, wz = w[:, 0], w[:, 1], w[:, 2]
    R = np.empty((aa.shape[0], 3, 3), dtype=np.float32)
    R[:, 0, 0] = c + wx * wx * (one - c); R[:, 1, 0] = wz * s + wx * wy * (one - c); R[:, 2, 0] = -wy * s + wx * wz * (one - c)
    R[:, 0, 1] = wx * wy * (one - c) - wz * s; R[:, 1, 1] = c + wy * wy * (one - c); R[:, 2, 1] = wx * s + wy * wz * (one - c)
    R[:, 0, 2] = wy * s + wx * wz * (one - c); R[:, 1, 2] = -wx * s + wy * wz * (one - c); R[:, 2, 2] = c + wz * wz * (one - c)
    small = theta2 <= np.float32(1e-6)
    if small.any():
        rx, ry, rz = aa[small, 0], aa[small, 1], aa[small, 2]
        T = np.zeros((small.sum(), 3, 3), dtype=np.float32)
        T[:, 0, 0] = T[:, 1, 1] = T[:, 2, 2] = 1
        T[:, 0, 1], T[:, 0, 2], T[:, 1, 0], T[:, 1, 2], T[:, 2, 0], T[:, 2, 1] = -rz, ry, rz, -rx, -ry, rx
        R[small] = T
    six = R[:, :, :2].transpose(0, 2, 1).reshape(-1, 6)
    return six.reshape(np.asarray(pose_aa).shape[0], -1, 6)


def process_qpos_list(qpos_list, amass_occlusion, target_fr=30, fix_height=None, log=print):
    """[(name, entry)] -> {"0-" + name: clip dict}; see the module docstring for the rules (process_amass_db.py:221-291)."""
    amass_res = {}
    for k, v in qpos_list:
        k = "0-" + k
        skip = int(v["mocap_framerate"] / target_fr)
        amass_pose = np.asarray(v["poses"])[::skip]
        amass_trans = np.asarray(v["trans"])[::skip]
        bound = amass_pose.shape[0]
        if k in amass_occlusion:
            issue = amass_occlusion[k]["issue"]
            if (issue == "sitting" or issue == "airborne") and "idxes" in amass_occlusion[k]:
                bound = amass_occlusion[k]["idxes"][0]  # annotated at 30 fps
                if bound < 10:
                    log("bound too small", k, bound)
                    continue
            else:
                log("issue irrecoverable", k, issue)
                continue
        if amass_pose.shape[0] < 10:
            continue
        pose_aa = amass_pose[:bound]
        trans = amass_trans[:bound]
        betas = np.asarray(v["betas"])
        fixed = fix_height is not None
        if fixed:
            trans = fix_height(pose_aa, betas, trans, v["gender"])
        amass_res[k] = {"pose_aa": pose_aa, "pose_6d": convert_aa_to_orth6d(pose_aa), "trans": trans, "beta": betas, "seq_name": k,
                        "gender": v["gender"], "height_fixed": fixed}
    return amass_res


def split_amass(amass_seq_data, log=print):
    """-> (train, test, valid) dictionaries by data-set prefix (process_amass_db.py:340-361, including its 'vald' / 'valid' slip)."""
    train, test, valid = {}, {}, {}
    for k, v in amass_seq_data.items():
        start_name = k.split("-")[1]
        found = False
        for dataset_key, split in AMASS_SPLIT_DICT.items():
            if start_name.lower().startswith(dataset_key.lower()):
                found = True
                if split == "test":
                    test[k] = v
                elif split == "valid":
                    valid[k] = v
                else:
                    train[k] = v
        if not found:
            log(f"Not found!! {start_name}")
    return train, test, valid


if __name__ == "__main__":
    import joblib
    ap = argparse.ArgumentParser()
    ap.add_argument("--amass_db", required=True, help="amass_db_smplx.pt (joblib)")
    ap.add_argument("--occlusion", default="sample_data/amass_copycat_occlusion_v2.pkl")
    ap.add_argument("--take", default="copycat_take5")
    ap.add_argument("--out_dir", default="sample_data")
    ap.add_argument("--smpl_dir", default="data/smpl", help="SMPL model files (licensed): with them the height fix of the reference is applied")
    args = ap.parse_args()
    np.random.seed(0)
    qpos_list = list(joblib.load(args.amass_db).items())
    np.random.shuffle(qpos_list)
    fix = None
    try:
        from ..smpllib.smpl_robot import SMPLBody, make_fix_height
        body = SMPLBody(args.smpl_dir)
        body._load(0)
        fix = make_fix_height(body)
    except (FileNotFoundError, ImportError) as e:
        print(f"no height fix ({e})")
    train, test, valid = split_amass(process_qpos_list(qpos_list, joblib.load(args.occlusion), fix_height=fix))
    for name, d in (("train", train), ("test", test), ("valid", valid)):
        joblib.dump(d, f"{args.out_dir}/amass_{args.take}_{name}.pkl")
