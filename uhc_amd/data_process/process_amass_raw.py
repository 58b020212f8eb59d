"""Raw AMASS download -> one joblib database (reference: uhc/data_process/process_amass_raw.py:83-131, 164-178).

Directory layout <dir>/<data set>/<subject>/<action>.npz; every action file (except the per-subject `shape.npz`) becomes the
entry "<data set>_<subject>_<action>" holding the file's arrays unchanged (betas, dmpls, gender, mocap_framerate, poses, trans).
The result is what `process_amass_db.py` reads.  Like the reference, a data set folder that contains plain files next to the
subject folders raises (os.listdir on a file)."""
import argparse
import os
import os.path as osp

import numpy as np

ALL_SEQUENCES = ["ACCAD", "BMLmovi", "BioMotionLab_NTroje", "CMU", "DFaust_67", "EKUT", "Eyes_Japan_Dataset", "HumanEva", "KIT", "MPI_HDM05",
                 "MPI_Limits", "MPI_mosh", "SFU", "SSM_synced", "TCD_handMocap", "TotalCapture", "Transitions_mocap", "BMLhandball", "DanceDB"]


def read_single_sequence(folder, seq_name):
    datas = {}
    for subject in os.listdir(folder):
        for action in [x for x in os.listdir(osp.join(folder, subject)) if x.endswith(".npz")]:
            fname = osp.join(folder, subject, action)
            if fname.endswith("shape.npz"):
                continue
            datas[f"{seq_name}_{subject}_{action[:-4]}"] = dict(np.load(fname))
    return datas


def read_data(folder, sequences="all", log=print):
    if sequences == "all":
        sequences = ALL_SEQUENCES
    db = {}
    for seq_name in sequences:
        datas = read_single_sequence(osp.join(folder, seq_name), seq_name)
        db.update(datas)
        log(seq_name, "number of seqs", len(datas))
    return db


if __name__ == "__main__":
    import joblib
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", type=str, default="data/amass", help="dataset directory")
    ap.add_argument("--out_dir", type=str, default="out")
    args = ap.parse_args()
    db = read_data(args.dir, sequences=ALL_SEQUENCES)
    joblib.dump(db, osp.join(args.out_dir, "amass_db_smplh.pt"))
