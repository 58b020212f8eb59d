"""AMASS clip loader with the reference's interface (uhc/data_loaders/dataset_amass_single.py:27-317):
``DatasetAMASSSingle(data_specs, data_mode)``, ``sample_seq``, ``get_sample_from_key``, ``iter_seq``,
``data_keys``, ``curr_key``, ``fr_start``, ``get_len``.  Input: joblib/pickle dict
{seq_name: {pose_aa (T,72), pose_6d (T,144), trans (T,3), beta (10|16), gender, ...}} at 30 fps.
Random draws use the same global generators, in the same order, as the reference (python `random` for the
uniform key choice, `numpy.random` for everything else), so a seeded run picks the same windows."""
import os
import pickle
import random
from collections import defaultdict

import numpy as np

from ..utils.math_utils import ewma

_GENDER = {"neutral": 0, "male": 1, "female": 2}


def _load_pickle(path):
    try:
        import joblib
        return joblib.load(open(path, "rb"))
    except ImportError:  # pragma: no cover
        return pickle.load(open(path, "rb"))


class DatasetAMASSSingle:
    def __init__(self, data_specs, data_mode="train", pickle_data=None):
        np.random.seed(0)  # the reference reseeds here (dataset_amass_single.py:30-31)
        random.seed(0)
        self.data_specs = data_specs
        self.data_mode = data_mode
        self.data_root = data_specs["file_path"] if data_mode == "train" else data_specs["test_file_path"]
        self.name = str(self.data_root).split("/")[-1]
        self.t_min = data_specs.get("t_min", 90)
        self.t_max = data_specs.get("t_max", -1)
        self.mode = data_specs.get("mode", "all")
        self.adaptive_iter = data_specs.get("adaptive_iter", -1)
        self.netural_path = data_specs.get("neutral_path", "sample_data/standing_neutral.pkl")
        self.pickle_data = pickle_data if pickle_data is not None else _load_pickle(self.data_root)
        self.netural_data = _load_pickle(self.netural_path) if os.path.exists(self.netural_path) else None
        self.init_probs = None
        self.data = self.process_data_pickle(self.pickle_data)
        self.traj_dim = next(iter(self.data["pose_6d"].values())).shape[1]
        self.seq_len = len(self.data["pose_6d"])
        self.iter_keys, self.seq_counter, self.curr_key = {}, 0, ""
        self.fr_start = self.fr_end = 0

    def process_data_pickle(self, pk_data):
        self.sample_keys, self.data_keys = [], []
        out = defaultdict(dict)
        keys = pk_data.keys() if self.mode == "all" else self.data_specs["key_subsets"]
        for k in keys:
            v = pk_data[k]
            T = v["pose_aa"].shape[0]
            if T < self.t_min + 1:
                continue
            out["pose_6d"][k], out["pose_aa"][k] = v["pose_6d"], v["pose_aa"]
            if "qpos" in v:
                out["qpos"][k] = v["qpos"]
            out["trans"][k] = v["trans"] if v["trans"].shape[0] == T else v["qpos"][:, :3]
            beta = np.asarray(v["beta"])
            beta = np.repeat(beta[None], T, axis=0) if beta.shape[0] != T else beta
            if beta.shape[1] != 16:
                beta = np.concatenate([beta, np.zeros((T, 16 - beta.shape[1]))], axis=1)
            out["beta"][k] = beta
            gender = v["gender"].item() if isinstance(v["gender"], np.ndarray) else v["gender"]
            gender = gender.decode("utf-8") if isinstance(gender, bytes) else gender
            if gender not in _GENDER:
                raise ValueError(f"gender '{gender}' not supported")
            out["gender"][k] = np.repeat([_GENDER[gender]], T, axis=0)
            out["obj_pose"][k] = v["obj_pose"] if v.get("obj_pose") is not None else v["pose_aa"]
            for opt in ("obj_info", "v_template"):
                if opt in v:
                    out[opt][k] = np.array(v[opt])
            reps = T // self.t_max + 1 if self.t_max != -1 else 1
            self.sample_keys += [(k, [-1])] * reps
            self.data_keys.append(k)
        return out

    def sample_seq(self, full_sample=False, freq_dict=None, sampling_temp=0.2, sampling_freq=0.5, precision_mode=False):
        if freq_dict is None or len(freq_dict.keys()) != len(self.data_keys):
            self.curr_key = random.choice(self.sample_keys)[0]
        else:
            succ = np.array([ewma(np.array(freq_dict[k])[:, 0] == 1) if len(freq_dict[k]) > 0 else 0 for k in freq_dict.keys()], dtype=np.float64)  # (one-entry histories are numpy bools)
            p = np.exp(-succ / sampling_temp)
            p = p / p.sum()
            self.curr_key = np.random.choice(self.data_keys, p=p) if np.random.binomial(1, sampling_freq) else np.random.choice(self.data_keys)
        return self.get_sample_from_key(self.curr_key, full_sample=full_sample, precision_mode=precision_mode, freq_dict=freq_dict,
                                        sampling_freq=sampling_freq)

    def get_sample_from_key(self, take_key, full_sample=False, freq_dict=None, fr_start=-1, precision_mode=False, sampling_freq=0.75):
        if take_key not in self.data["pose_aa"]:
            raise KeyError("Key not found")
        self.curr_key = take_key
        T = self.data["pose_aa"][take_key].shape[0]
        if full_sample:
            fr_start, fr_end = 0, T
        else:
            if freq_dict is not None and precision_mode:
                perfs = np.array(freq_dict[take_key])
                failed = perfs[perfs[:, 0] != 1][:, 1] if len(perfs) > 0 else []
                if len(failed) > 0 and np.random.binomial(1, sampling_freq):
                    c = np.random.choice(failed)
                    fr_start = np.random.randint(max(c - 20 - self.t_min, 0), min(c + 20, T - self.t_min))
                else:
                    fr_start = np.random.randint(0, T - self.t_min)
            elif fr_start == -1:
                fr_start = np.random.randint(0, T - self.t_min)
            fr_end = fr_start + self.t_max if (self.t_max != -1 and fr_start + self.t_max < T) else T
        self.fr_start, self.fr_end = fr_start, fr_end
        sample = {k: d[take_key][fr_start:fr_end] for k, d in self.data.items() if k not in ("obj_info", "v_template")}
        n = self.data["pose_aa"][take_key][fr_start:fr_end]
        sample["seq_name"] = take_key
        sample["has_obj"] = sample["obj_pose"].shape != n.shape
        sample["num_obj"] = sample["obj_pose"].shape[1] // 7 if sample["has_obj"] else 0
        for opt in ("obj_info", "v_template"):
            if opt in self.data:
                sample[opt] = self.data[opt][take_key]
        return sample

    def sample_windows(self, n, freq_dict=None, sampling_temp=0.2, sampling_freq=0.5):
        """n draws of (key, fr_start, fr_end) with the distribution of n sample_seq() calls (non-precision mode), for
        the batched env: the success-weighted key probabilities are computed once instead of once per draw."""
        if freq_dict is None or len(freq_dict.keys()) != len(self.data_keys):
            keys = [random.choice(self.sample_keys)[0] for _ in range(n)]
        else:
            succ = np.array([ewma(np.array(freq_dict[k])[:, 0] == 1) if len(freq_dict[k]) > 0 else 0 for k in freq_dict.keys()], dtype=np.float64)  # (one-entry histories are numpy bools)
            p = np.exp(-succ / sampling_temp)
            p = p / p.sum()
            weighted = np.random.binomial(1, sampling_freq, size=n).astype(bool)
            idx = np.where(weighted, np.random.choice(len(self.data_keys), size=n, p=p), np.random.choice(len(self.data_keys), size=n))
            keys = [self.data_keys[i] for i in idx]
        T = np.array([self.data["pose_aa"][k].shape[0] for k in keys])
        fs = np.random.randint(0, T - self.t_min)
        fe = np.where((self.t_max != -1) & (fs + self.t_max < T), fs + self.t_max, T)
        if n:
            self.curr_key, self.fr_start, self.fr_end = keys[-1], int(fs[-1]), int(fe[-1])
        return keys, fs, fe

    def get_sample_len_from_key(self, take_key):
        return self.data["pose_aa"][take_key].shape[0]

    def set_singles(self, seq_name):
        self.data_keys = [seq_name]

    def set_seq_counter(self, idx):
        self.seq_counter = idx

    def iter_seq(self):
        self.iter_keys = self.data_keys
        self.curr_key = self.iter_keys[self.seq_counter % len(self.iter_keys)]
        self.seq_counter += 1
        return self.get_sample_from_key(self.curr_key, full_sample=True, fr_start=0)

    def get_len(self):
        return len(self.data_keys)
