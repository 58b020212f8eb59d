"""How far do the least certain restatements of MuJoCo move a trajectory?  ("next" 7.)

MuJoCo 2.1.0 itself cannot be had here (closed binary, absent from the reference tree and the image), so three [MJ-ext] choices of
oracle/physics_oracle.c are unpinned ("soft spots").  This tool runs the ORACLE ALONE (float64, exact contact solve,
the reference's control loop: stable PD towards the clip's next frame, zero policy action) for 200 control steps on the model class the
reference generates (body-body collisions on) and reports max |delta qpos| against the baseline for each alternative:

  maxcon3     plane-mesh multi-contact cap 3 instead of 4 (support vertex + hull-graph neighbours within the margin);
  qhull       hull graph recomputed by a convex-hull run (scipy / qhull triangulation) instead of the shipped STL triangles' edges:
              the same hull, another triangulation of its flat faces -> other neighbour lists, other neighbour ORDER;
  hillclimb   MPR support vertex by hill-climbing on the hull graph (cached start vertex) instead of the exhaustive first-maximum scan.

A future comparison against a real MuJoCo (tests/test_mujoco_live.py) then knows which of the three to look at first.

    python tools/sensitivity.py > profiles/r04_sensitivity.txt
"""
import dataclasses
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def qhull_graph(model):
    """The model with every hull's adjacency rebuilt from scipy.spatial.ConvexHull(vertices).simplices."""
    from scipy.spatial import ConvexHull
    from uhc_amd.model.mjcf import hull_adjacency
    o = model.copy()
    adr, adj = [0], []
    for g in range(model.ngeom):
        va, vn = int(model.geom_vertadr[g]), int(model.geom_vertnum[g])
        if vn == 0:
            continue
        h = ConvexHull(np.asarray(model.mesh_vert[va:va + vn]))
        a, idx = hull_adjacency(vn, np.asarray(h.simplices))
        for v in range(vn):
            adj += [int(i) + va for i in idx[a[v]:a[v + 1]]]
            adr.append(len(adj))
    o.mesh_adjadr = np.array(adr, dtype=np.int32)
    o.mesh_adj = np.array(adj, dtype=np.int32)
    o.nmeshadj = len(adj)
    return o


def run(model, ctrl, qpos_clip, qvel0, steps, support_mode=0):
    from oracle.physics import OracleSim, lib
    lib().orc_set_support_mode(int(support_mode))
    o = OracleSim(model, ctrl)
    o.set_state(qpos_clip[0], qvel0)
    act = np.zeros(ctrl.action_dim)
    traj, ncon = [o.get("qpos").copy()], []
    for t in range(steps):
        o.do_simulation(act, qpos_clip[min(t + 1, len(qpos_clip) - 1)][7:])
        traj.append(o.get("qpos").copy())
        ncon.append(o.geti("ncon"))
    lib().orc_set_support_mode(0)
    return np.array(traj), np.array(ncon)


def main():
    import torch
    from uhc_amd import sim as S
    from uhc_amd.data_loaders.synthetic import make_synthetic_amass
    from uhc_amd.smpllib.smpl_mujoco import smpl_to_qpose
    from uhc_amd.smpllib.smpl_robot import robot_variant
    from uhc_amd.smpllib.torch_smpl_humanoid import Humanoid
    base = dataclasses.replace(robot_variant(S.load_asset_model(), {"mesh": True, "model": "smpl"}), solver=1, iterations=300)
    ctrl = S.make_ctrl(base)
    stand = np.load(os.path.join(ROOT, "uhc_amd", "assets", "standing_neutral.npz"))
    clips = {"standing_neutral (187 frames of the shipped standing pose)": np.tile(np.asarray(stand["qpos"]).reshape(1, -1), (187, 1))}
    syn = make_synthetic_amass(1, seed=3, t_range=(220, 220), amp=0.3)
    c = next(iter(syn.values()))
    clips["synthetic perturbed-standing clip (seed 3, 220 frames, joint amplitudes <= 0.3 rad)"] = smpl_to_qpose(c["pose_aa"], base, trans=c["trans"], count_offset=True)
    variants = {"maxcon3": (dataclasses.replace(base, plane_mesh_maxcon=3), 0), "qhull": (qhull_graph(base), 0), "hillclimb": (base, 1)}
    steps = 200
    marks = [1, 5, 10, 20, 30, 60, 100, 150, 200]
    print(__doc__.split("\n\n")[0])
    print(f"model: generated class (self-collision on, rel_joint_lm), plane_mesh_maxcon {base.plane_mesh_maxcon}, exact contact solve; {steps} control steps of 15 substeps; zero policy action\n")
    g = qhull_graph(base)
    same = sum(set(base.mesh_adj[base.mesh_adjadr[v]:base.mesh_adjadr[v + 1]]) == set(g.mesh_adj[g.mesh_adjadr[v]:g.mesh_adjadr[v + 1]]) for v in range(base.nmeshvert))
    print(f"hull graph: {same} of {base.nmeshvert} vertices have the same neighbour SET under both triangulations ({base.nmeshadj} vs {g.nmeshadj} directed edges)\n")
    for name, qp in clips.items():
        feat = Humanoid(model=base).qpos_fk(torch.from_numpy(np.asarray(qp)[:2].copy()))
        ref, ncon = run(base, ctrl, qp, feat["qvel"][0], steps)
        print(f"clip: {name}; baseline root height after {marks}: " + " ".join(f"{ref[m, 2]:.3f}" for m in marks) + f"; contacts mean {ncon.mean():.1f} max {ncon.max()}")
        print(f"  {'variant':10s} " + " ".join(f"{'t=' + str(m):>9s}" for m in marks) + "   first step with max |dq| > 1e-9 / > 1e-4")
        for vn, (vm, mode) in variants.items():
            tr, _ = run(vm, ctrl, qp, feat["qvel"][0], steps, support_mode=mode)
            d = np.abs(tr - ref).max(axis=1)
            f9 = int(np.argmax(d > 1e-9)) if (d > 1e-9).any() else -1
            f4 = int(np.argmax(d > 1e-4)) if (d > 1e-4).any() else -1
            print(f"  {vn:10s} " + " ".join(f"{d[m]:9.2e}" for m in marks) + f"   {f9} / {f4}")
        print()
    print("reading: a column is max over all 76 coordinates of |qpos(variant) - qpos(baseline)| at that control step (30 steps = 1 s).  Where a\n"
          "variant first departs (> 1e-9) is where its code path first mattered; once apart, contact-rich motion amplifies any difference (the\n"
          "humanoid under zero policy action falls within ~1 s: DESIGN.md section 2), so late columns measure chaos, the FIRST-departure step\n"
          "and the early columns measure the restatement.")


if __name__ == "__main__":
    main()
