"""SMPL shape -> humanoid model: the reference's per-episode model generator (uhc/smpllib/smpl_robot.py:83-147 `get_joint_geometries`,
:1018-1257 `Robot.load_from_skeleton`; uhc/smpllib/smpl_parser.py:386-424 `get_mesh_offsets`; uhc/khrylib/mocap/skeleton_mesh.py:131-324
MJCF writer), re-built around in-memory data:

    (beta, gender) --SMPL forward--> verts (6890, 3), joints (24, 3), skin weights (6890, 24)
        -> per joint: the vertices it owns (arg-max skin weight), scaled about the joint, convex hull, decimated to ~50 vertices
        -> kinematic tree in SMPL_BONE_KINTREE order, three hinges (z, y, x) or one ball joint per bone, hull meshes, motors,
           body-body collisions on (contype = conaffinity = 1) with Chest / shoulders excluded
        -> MJCF string (export_xml_string) and the compiled Model (get_model) with the meshes handed over in memory.

The reference writes STL files to /tmp, decimates them with VTK's quadric decimation and lets MuJoCo compile the XML; here nothing
touches the disk, the decimation is a greedy least-volume-loss vertex removal on the hull (same target: min_num_vert = 50), and the
model compiler is uhc_amd/model/mjcf.py.  The SMPL forward pass needs the licensed SMPL model files (not shipped, absent from the
reference too: `data/smpl`); `SMPLBody` loads them when present and raises a clear error otherwise.  Everything after it runs on any
(verts, joints, skin weights) triple, which is what the tests feed."""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional

import numpy as np

from .smpl_mujoco import SMPL_BONE_KINTREE_NAMES, SMPL_BONE_ORDER_NAMES

SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]  # the SMPL kinematic tree (joint order)


class SMPLBody:
    """SMPL forward pass at the zero pose: v = v_template + shapedirs beta, J = J_regressor v (smpl_parser.py:386-424 needs no pose
    blend shapes: `get_mesh_offsets` evaluates the zero pose).  Model data: an .npz / .pkl with v_template, shapedirs, J_regressor,
    weights under `data_dir` (SMPL_NEUTRAL / SMPL_MALE / SMPL_FEMALE), as distributed by the SMPL authors (licensed)."""

    FILES = {0: "SMPL_NEUTRAL", 1: "SMPL_MALE", 2: "SMPL_FEMALE"}

    def __init__(self, data_dir="data/smpl"):
        self.data_dir = data_dir
        self._cache: Dict[int, dict] = {}

    def _load(self, gender: int) -> dict:
        if gender in self._cache:
            return self._cache[gender]
        stem = os.path.join(self.data_dir, self.FILES[int(gender)])
        for ext in (".npz", ".pkl"):
            if os.path.exists(stem + ext):
                if ext == ".npz":
                    z = dict(np.load(stem + ext, allow_pickle=True))
                else:
                    import pickle
                    z = pickle.load(open(stem + ext, "rb"), encoding="latin1")
                d = {k: np.asarray(z[k].todense() if hasattr(z[k], "todense") else z[k], dtype=np.float64) for k in ("v_template", "shapedirs", "J_regressor", "weights")}
                self._cache[gender] = d
                return d
        raise FileNotFoundError(f"SMPL model file {stem}.npz|.pkl not found: the SMPL body models are licensed and not shipped; place them under "
                                f"{self.data_dir} (the reference expects them under data/smpl as well)")

    def __call__(self, betas, gender: int = 0):
        d = self._load(gender)
        b = np.asarray(betas, dtype=np.float64).reshape(-1)
        nb = min(len(b), d["shapedirs"].shape[-1])
        verts = d["v_template"] + d["shapedirs"][..., :nb] @ b[:nb]
        joints = d["J_regressor"] @ verts
        return verts, joints, d["weights"]


def in_hull(hull, queries, tolerance=1e-3):
    """Which query points lie inside (or within `tolerance` of) a scipy ConvexHull (smpl_robot.py:73-80): every facet's plane equation
    n . x + d evaluated at the point must be <= tolerance."""
    q = np.atleast_2d(np.asarray(queries, dtype=np.float64))
    return (q @ hull.equations[:, :-1].T + hull.equations[:, -1] <= tolerance).all(axis=1)


def pose_body(verts, joints, skin_weights, pose_aa):
    """Linear blend skinning of a rest-pose body (what `body_provider` returns) by 24 axis-angle joint rotations on the SMPL tree:
    v' = sum_j w_j G_j [v; 1], G_j = prod over the chain of [R_k | J_k - R_k J_k] (smplx's lbs without the pose blend shapes, whose
    effect on the lowest vertex of a standing or walking frame is millimetres).  Returns the posed vertices, root at its rest position."""
    from scipy.spatial.transform import Rotation as sRot
    R = sRot.from_rotvec(np.asarray(pose_aa, dtype=np.float64).reshape(24, 3)).as_matrix()
    J = np.asarray(joints, dtype=np.float64)
    G = np.zeros((24, 4, 4))
    for j in range(24):
        L = np.eye(4)
        L[:3, :3] = R[j]
        L[:3, 3] = J[j] - (J[SMPL_PARENTS[j]] if j > 0 else 0.0)
        G[j] = L if j == 0 else G[SMPL_PARENTS[j]] @ L
    for j in range(24):  # to transforms of rest-pose points: subtract the rest joint position first
        G[j][:3, 3] -= G[j][:3, :3] @ J[j]
    v = np.asarray(verts, dtype=np.float64)
    T = np.einsum("vj,jab->vab", np.asarray(skin_weights, dtype=np.float64), G)
    return np.einsum("vab,vb->va", T[:, :3, :3], v) + T[:, :3, 3]


def make_fix_height(body_provider: Callable):
    """fix_height_smpl_vanilla (uhc/data_process/process_amass_db.py:194-219): shift the clip so that the lowest vertex of its FIRST frame
    touches z = 0.  -> callable (pose_aa (T, 72), betas, trans (T, 3), gender) -> trans, the `fix_height=` hook of process_qpos_list.
    body_provider: SMPLBody (the licensed files) or any (betas, gender) -> (vertices, joints, skin weights)."""
    names = {"neutral": 0, "male": 1, "female": 2}

    def fix(pose_aa, betas, trans, gender):
        g = gender.item() if isinstance(gender, np.ndarray) else gender
        g = g.decode("utf-8") if isinstance(g, bytes) else g
        if g not in names and g not in (0, 1, 2):
            raise Exception("Gender Not Supported!!")
        verts, joints, W = body_provider(np.asarray(betas, dtype=np.float64).reshape(-1)[:10], names.get(g, g))
        v0 = pose_body(verts, joints, W, np.asarray(pose_aa)[0]) + np.asarray(trans, dtype=np.float64)[0]
        out = np.array(trans, dtype=np.float64, copy=True)
        out[:, 2] -= v0[:, 2].min()
        return out

    return fix


# --------------------------------------------------------------------------------------------------------------------- hulls
def _outward(tris: np.ndarray, centre: np.ndarray) -> np.ndarray:
    n = np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0])
    flip = np.einsum("ij,ij->i", n, tris.mean(axis=1) - centre) < 0  # smpl_robot.py:123-131: flip faces whose normal points inwards
    tris[flip] = tris[flip][:, [0, 2, 1]]
    return tris


def decimate_hull(points: np.ndarray, target: int) -> np.ndarray:
    """Hull vertices of `points`, thinned to at most `target` of them: repeatedly drop the hull vertices whose removal loses the least
    volume -- the cap a vertex cuts off is ~ (its height above the plane of its neighbour ring) x (the ring's area) / 3 -- and re-hull,
    until the count fits.  Stands in for the reference's VTK quadric decimation to min_num_vert = 50 (smpl_robot.py:140-145): MuJoCo
    re-hulls the decimated mesh anyway, so only the vertex budget and the closeness to the original hull matter."""
    from scipy.spatial import ConvexHull
    pts = np.asarray(points, dtype=np.float64)
    pts = pts[ConvexHull(pts).vertices]
    while len(pts) > target:
        hull = ConvexHull(pts)
        nbr: List[set] = [set() for _ in range(len(pts))]
        for f in hull.simplices:
            for a_ in f:
                nbr[a_].update(int(x) for x in f if x != a_)
        loss = np.full(len(pts), np.inf)
        for v in hull.vertices:
            ring = pts[sorted(nbr[v])]
            if len(ring) < 3:
                continue
            c = ring.mean(axis=0)
            _, _, vt = np.linalg.svd(ring - c)
            uv = (ring - c) @ vt[:2].T                      # ring in its least-squares plane
            order = np.argsort(np.arctan2(uv[:, 1], uv[:, 0]))
            x, y = uv[order, 0], uv[order, 1]
            area = 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))
            loss[v] = abs((pts[v] - c) @ vt[2]) * area / 3.0
        k = max(1, min(len(pts) - target, len(pts) // 10))  # a few per re-hull while far from the target (never two neighbours at once)
        drop, taken = [], set()
        for v in np.argsort(loss):
            if len(drop) == k:
                break
            if v in taken:
                continue
            drop.append(int(v))
            taken.update(nbr[v])
        keep = np.ones(len(pts), dtype=bool)
        keep[drop] = False
        if keep.sum() < 4:
            break
        pts = pts[keep]
    return pts


def get_joint_geometries(smpl_verts, smpl_jts, skin_weights, joint_names, scale_dict=None, min_num_vert=50):
    """Per joint: the SMPL vertices whose largest skin weight is this joint's, scaled about the joint, their convex hull as outward
    triangles, decimated to ~min_num_vert vertices (smpl_robot.py:83-147).  Returns {joint name: (ntri, 3, 3) triangles in the SMPL frame}."""
    from scipy.spatial import ConvexHull
    scale_dict = scale_dict or {}
    vert_to_joint = np.asarray(skin_weights).argmax(axis=1)
    out = {}
    for jind, jname in enumerate(joint_names):
        vind = np.where(vert_to_joint == jind)[0]
        if len(vind) == 0:
            continue
        vert = (smpl_verts[vind] - smpl_jts[jind]) * scale_dict.get(jname, 1) + smpl_jts[jind]
        hv = decimate_hull(vert, max(min_num_vert, 4))
        hull = ConvexHull(hv)
        out[jname] = _outward(hv[hull.simplices].copy(), hv[hull.vertices].mean(axis=0))
    return out


# --------------------------------------------------------------------------------------------------------------------- MJCF
TEMPLATE = """<mujoco model="humanoid">
  <compiler angle="degree" inertiafromgeom="true" coordinate="global"/>
  <default>
    <joint limited="true" armature="0.01"/>
    <geom conaffinity="1" condim="1" contype="1" margin="0.001" rgba="0.8 0.6 .4 1"/>
    <motor ctrllimited="true" ctrlrange="-1 1"/>
  </default>
  <option timestep="0.00222222222"/>
  <asset>{assets}</asset>
  <worldbody>
    <geom condim="3" friction="1. .1 .1" name="floor" pos="0 0 0" size="100 100 0.2" type="plane" conaffinity="1" contype="1"/>
{bodies}  </worldbody>
  <contact>{excludes}</contact>
  <actuator>{motors}</actuator>
</mujoco>
"""


def default_joint_range(joint_names, rel_joint_lm=True):
    """smpl_parser.py:320-325 (+-180 deg, elbows +-720) with the tight knee / ankle / toe ranges of rel_joint_lm (smpl_robot.py:1087-1110), degrees."""
    r = {n: np.tile(np.array([-180.0, 180.0]), (3, 1)) for n in joint_names}
    for n in ("L_Elbow", "R_Elbow"):
        if n in r:
            r[n] = np.tile(np.array([-720.0, 720.0]), (3, 1))
    if rel_joint_lm:
        d = np.rad2deg
        for s in ("L_", "R_"):
            if s + "Knee" in r:
                r[s + "Knee"] = np.array([[-d(np.pi / 16), d(np.pi / 16)], [-d(np.pi / 16), d(np.pi / 16)], [-d(np.pi / 16), 180.0]])
            if s + "Ankle" in r:
                r[s + "Ankle"] = np.tile(np.array([-90.0, 90.0]), (3, 1))
            if s + "Toe" in r:
                r[s + "Toe"] = np.array([[-45.0, 45.0], [-45.0, 45.0], [-90.0, 90.0]])
    return r


def robot_variant(model, robot_cfg: dict):
    """The shipped static asset turned into the model `Robot(robot_cfg)` GENERATES for the same body (what the reference's env runs on:
    HumanoidEnv.__init__ builds its MuJoCo model from `cfg.robot_cfg`, never from the asset file, uhc/envs/humanoid_im.py:52-64):
      * body-body collisions on, Chest excluded against both shoulders (smpl_parser.py:327-328, smpl_robot.py:1177-1198) -- unless the
        config says `self_collision: false` (this build's own key: the floor-only behaviour of the static asset);
      * `rel_joint_lm` (default true): the tight knee / ankle / toe ranges (smpl_robot.py:1087-1110);
      * `ball: true`: one ball joint + three gear-vector motors per bone (skeleton_mesh_v2.py:183-193, :243-267).
    Used where the SMPL model files (licensed, absent here) are not available to run the generator itself."""
    from ..model.mjcf import JNT_HINGE, ball_variant, self_collision_variant, set_const
    cfg = dict(robot_cfg or {})
    m = model
    if cfg.get("mesh", True) and cfg.get("self_collision", True):
        m = self_collision_variant(m)
    if cfg.get("rel_joint_lm", True) and not cfg.get("ball", False):
        rng = default_joint_range([n for n in m.body_names[1:]], True)
        m = m.copy()
        for j, name in enumerate(m.joint_names):
            if m.jnt_type[j] != JNT_HINGE or "_" not in name:
                continue
            body, ax = name.rsplit("_", 1)
            if body in rng and ax in ("z", "y", "x"):
                m.jnt_range[j] = np.deg2rad(rng[body]["zyx".index(ax)])
    if cfg.get("ball", False):
        m = ball_variant(m)
    return m


class Robot:
    """`Robot(cfg).load_from_skeleton(betas, gender)` -> `export_xml_string()` / `get_model()` (smpl_robot.py:917-1016, :1018-1257).
    cfg: the `robot` block of a config (mesh, ball, flatfoot, rel_joint_lm, model).  body_provider: callable (betas, gender) ->
    (verts, joints, skin_weights); default = SMPLBody(data_dir)."""

    def __init__(self, cfg: dict, data_dir: str = "data/smpl", body_provider: Optional[Callable] = None):
        self.cfg = dict(cfg or {})
        self.mesh = self.cfg.get("mesh", True)
        self.ball = self.cfg.get("ball", False)
        self.flatfoot = self.cfg.get("flatfoot", False)
        self.rel_joint_lm = self.cfg.get("rel_joint_lm", True)
        self.smpl_model = self.cfg.get("model", "smpl")
        self.masterfoot = bool(self.cfg.get("masterfoot", False))  # twelve capsule bodies under each ankle (smpl_robot.py:1174-1175, :1336-1414)
        self.master_range = float(self.cfg.get("master_range", 30))
        if not self.mesh:
            raise NotImplementedError("capsule (non-mesh) humanoids are outside the hot path: every copycat / smpl_shape config sets mesh: True")
        self.body_provider = body_provider or SMPLBody(data_dir)
        self.joint_names = list(SMPL_BONE_ORDER_NAMES)
        self.xml, self.meshes, self.beta, self.height, self.bone_length = None, None, None, None, None

    def load_from_skeleton(self, betas=None, v_template=None, gender=(0,), objs_info=None, obj_pose=None, params=None):
        if objs_info is not None:
            raise NotImplementedError("objects are added to a compiled model with uhc_amd.model.mjcf.add_free_bodies")
        g = int(np.asarray(gender).reshape(-1)[0])
        if g not in (0, 1, 2):
            raise Exception("Gender Not Supported!!")
        b = np.zeros(10) if betas is None else np.asarray(betas, dtype=np.float64).reshape(-1)
        if self.smpl_model == "smpl" and len(b) == 16:  # smpl_robot.py:1056-1058
            b = b[:10]
        self.beta = b
        verts, joints, skin_weights = self.body_provider(b, g)
        verts = np.array(verts, dtype=np.float64)
        if self.flatfoot:  # smpl_parser.py:394-396: flatten the soles
            feet = verts[:, 1] < verts[:, 1].min() + 0.01
            verts[feet, 1] = verts[feet][:, 1].mean()
        joints = np.asarray(joints, dtype=np.float64)
        self.height = float(verts[:, 1].max() - verts[:, 1].min())
        offsets = {n: (joints[c] - joints[p]) if c > 0 else joints[c] for c, (n, p) in enumerate(zip(self.joint_names, SMPL_PARENTS))}
        self.bone_length = np.array([np.linalg.norm(v) for v in offsets.values()])
        self.meshes = get_joint_geometries(verts, joints, skin_weights, self.joint_names)
        self.xml = self._write_xml(joints, default_joint_range(self.joint_names, self.rel_joint_lm))
        return self

    def _write_xml(self, joints, joint_range):
        pos = {n: joints[i] for i, n in enumerate(self.joint_names)}
        parent = {n: (self.joint_names[p] if p >= 0 else None) for n, p in zip(self.joint_names, SMPL_PARENTS)}
        children = {n: [c for c in SMPL_BONE_KINTREE_NAMES if parent[c] == n] for n in self.joint_names}
        motors = []

        def body(name, depth):
            ind = "  " * (depth + 2)
            p = pos[name]
            pstr = f"{p[0]:.4f} {p[1]:.4f} {p[2]:.4f}"
            s = f'{ind}<body name="{name}" pos="{pstr}">\n'
            if parent[name] is None:
                s += f'{ind}  <joint name="{name}" pos="{pstr}" limited="false" type="free" armature="0" damping="0" stiffness="0" frictionloss="0"/>\n'
            elif self.ball:
                s += f'{ind}  <joint name="{name}" type="ball" pos="{pstr}" limited="false"/>\n'
                for k, ch in enumerate("zyx"):
                    ax = ["0 0 1", "0 1 0", "1 0 0"][k]
                    motors.append(f'<motor name="{name}_{ch}" joint="{name}" gear="{ax}"/>')
            else:
                for k, ch in enumerate("zyx"):  # skeleton_mesh.py:243-257: three hinges <Body>_z, _y, _x at the body origin
                    ax = ["0 0 1", "0 1 0", "1 0 0"][k]
                    lo, hi = joint_range[name][k]
                    s += (f'{ind}  <joint name="{name}_{ch}" type="hinge" pos="{pstr}" axis="{ax}" stiffness="0" damping="0" armature="0.01" '
                          f'range="{lo:.4f} {hi:.4f}"/>\n')
                    motors.append(f'<motor name="{name}_{ch}" joint="{name}_{ch}" gear="1"/>')
            if name in self.meshes:
                s += f'{ind}  <geom type="mesh" mesh="{name}" contype="1" conaffinity="1"/>\n'
            for c in children[name]:
                s += body(c, depth + 1)
            if self.masterfoot and name in ("L_Ankle", "R_Ankle"):
                s += self._masterfoot_bodies(name, pos, children, depth + 1, motors)
            return s + f"{ind}</body>\n"

        bodies = body(self.joint_names[0], 0)
        assets = "".join(f'<mesh name="{n}" file="{n}.stl"/>' for n in self.meshes)
        ex = '<exclude name="add01" body1="L_Shoulder" body2="Chest"/><exclude name="add02" body1="R_Shoulder" body2="Chest"/>'  # smpl_robot.py:1177-1198
        return TEMPLATE.format(assets=assets, bodies=bodies, excludes=ex, motors="".join(motors))

    # the reference's template of the twelve toe capsules of one foot, before scaling (smpl_robot.py:1343-1356)
    MASTERFOOT_TEMPLATE = np.array([[0, -0.15, 0], [-0.08, -0.15, 0.1], [0.08, -0.15, 0.1], [-0.1, -0.15, 0.2], [0.1, -0.15, 0.2], [-0.1, -0.15, 0.35], [0.1, -0.15, 0.35],
                                    [-0.1, -0.17, 0.6], [0.1, -0.17, 0.6], [0, -0.17, 0.6], [0.05, -0.17, 0.6], [-0.05, -0.17, 0.6]])

    def _masterfoot_bodies(self, ankle, pos, children, depth, motors):
        """`Robot.add_masterfoot` (uhc/smpllib/smpl_robot.py:1336-1414): twelve child bodies per ankle, each a CLONE of the ankle's body node -- its origin and its three
        hinges at the ankle joint -- with the hull replaced by one capsule (radius 0.035, 0.1 long along x, contype 0 / conaffinity 1: the toes meet the floor and the
        other bodies' hulls, not each other), the hinges limited to +-master_range degrees and one motor per hinge.  The capsules' start points follow the template
        scaled by the foot's length (ankle-to-toe distance over 0.1343...) and sit at the height of the ankle hull's lowest vertex.  49 bodies / 147 dofs with both
        feet -- beyond what the HIP step kernels hold (nv <= 128: uhc_batch_create refuses the model and says so); model compiler and oracle take it."""
        ind = "  " * (depth + 2)
        p = pos[ankle]
        toe = children[ankle][0]
        diff_mul = float(np.linalg.norm(p - pos[toe]) / 0.13432456960660616)
        t = self.MASTERFOOT_TEMPLATE.copy()
        t[:, 2] -= 0.08 * diff_mul
        t[:, 0] -= 0.05 * diff_mul if ankle == "R_Ankle" else -0.05 * diff_mul
        t /= 3 / diff_mul
        t += p
        t[:, 1] = float(self.meshes[ankle].reshape(-1, 3)[:, 1].min())
        pstr = f"{p[0]:.4f} {p[1]:.4f} {p[2]:.4f}"
        fmt = lambda x: f"{x:.6f}".rstrip("0").rstrip(".")
        out = ""
        for i, a in enumerate(t):
            name = f"{ankle}_master{i}"
            out += f'{ind}<body name="{name}" pos="{pstr}">\n'
            for k, ch in enumerate("zyx"):
                ax = ["0 0 1", "0 1 0", "1 0 0"][k]
                out += (f'{ind}  <joint name="{name}_{ch}" type="hinge" pos="{pstr}" axis="{ax}" stiffness="0" damping="0" armature="0.01" '
                        f'range="-{self.master_range:g} {self.master_range:g}"/>\n')
                motors.append(f'<motor name="{name}_{ch}" joint="{name}_{ch}" gear="1"/>')
            out += (f'{ind}  <geom type="capsule" size="0.035" fromto="{fmt(a[0])} {fmt(a[1])} {fmt(a[2])} {fmt(a[0] + 0.1)} {fmt(a[1])} {fmt(a[2])}" '
                    f'contype="0" conaffinity="1"/>\n')
            out += f"{ind}</body>\n"
        return out

    def export_xml_string(self) -> bytes:
        return self.xml.encode("utf-8")

    def get_model(self):
        """The compiled model of the current shape (what reload_sim_model + load_model_from_xml produce, humanoid_im.py:1441-1454)."""
        from ..model.mjcf import compile_mjcf
        return compile_mjcf(self.xml, meshes=self.meshes)


def default_body_provider(cfg):
    """SMPLBody over <base_dir>/data/smpl when the (licensed) SMPL model files are there, else None."""
    import os
    d = os.path.join(getattr(cfg, "base_dir", "."), "data", "smpl")
    if all(os.path.exists(os.path.join(d, f)) for f in ("SMPL_NEUTRAL.pkl", "SMPL_MALE.pkl", "SMPL_FEMALE.pkl")):
        return SMPLBody(d)
    return None


def generate_shape_models(robot_cfg: dict, clips: dict, body_provider: Callable, solver_fields: Optional[dict] = None):
    """What the reference does at every load_expert -- reset_robot: Robot.load_from_skeleton(beta, gender) -> export_xml_string ->
    reload_sim_model (uhc/envs/humanoid_im.py:154-190, :1441-1454) -- done once for every distinct (beta, gender) among the clips.
    clips: {key: {"beta": (10|16,) or (T, 16), "gender": ...}}.  Returns (models, clip_model): compiled models brought to one mesh layout
    (`common_mesh_layout`: they share a batch, one blob each) and {key: index into models}."""
    from ..model.mjcf import common_mesh_layout
    robot = Robot(robot_cfg, body_provider=body_provider)
    seen, models, clip_model = {}, [], {}
    for k, c in clips.items():
        beta = np.asarray(c["beta"], dtype=np.float64)
        beta = beta[0] if beta.ndim == 2 else beta
        g = int(np.asarray(c["gender"]).reshape(-1)[0])
        tag = (beta.round(6).tobytes(), g)
        if tag not in seen:
            robot.load_from_skeleton(beta, gender=[g])
            seen[tag] = len(models)
            models.append(robot.get_model())
        clip_model[k] = seen[tag]
    return common_mesh_layout(models), clip_model
