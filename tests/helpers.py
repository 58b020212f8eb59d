"""Shared builders for tests (synthetic MJCF models written for this repo)."""
import numpy as np

BOX_TRIS = None


def box_triangles(hx, hy, hz, center=(0, 0, 0)):
    c = np.array(center, dtype=np.float64)
    v = np.array([[sx * hx, sy * hy, sz * hz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64) + c
    # vertex index = 4*ix + 2*iy + iz
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    tris = []
    for a, b, cc, d in quads:
        tris.append([v[a], v[b], v[cc]])
        tris.append([v[a], v[cc], v[d]])
    return np.array(tris)


PENDULUM_XML = """
<mujoco>
  <compiler angle="radian" coordinate="local" inertiafromgeom="true"/>
  <option timestep="0.0005"/>
  <default><geom contype="0" conaffinity="0" margin="0.001"/></default>
  <asset><mesh name="bob" file="unused.stl"/></asset>
  <worldbody>
    <body name="arm" pos="0 0 2">
      <joint name="hinge" type="hinge" axis="0 1 0" pos="0 0 0"/>
      <geom type="mesh" mesh="bob"/>
    </body>
  </worldbody>
  <actuator><motor name="m" joint="hinge" gear="1"/></actuator>
</mujoco>
"""

BOX_ON_PLANE_XML = """
<mujoco>
  <compiler angle="radian" coordinate="local" inertiafromgeom="true"/>
  <option timestep="0.002"/>
  <default><geom contype="1" conaffinity="1" condim="3" margin="0.001"/></default>
  <asset><mesh name="box" file="unused.stl"/></asset>
  <worldbody>
    <geom name="floor" type="plane" size="10 10 0.1" pos="0 0 0"/>
    <body name="box" pos="0 0 0.1">
      <joint name="root" type="free"/>
      <geom type="mesh" mesh="box" condim="1"/>
    </body>
  </worldbody>
</mujoco>
"""


def pendulum_model(length=0.5, half=0.05):
    from uhc_amd.model.mjcf import compile_mjcf
    tris = box_triangles(half, half, half, center=(0, 0, -length))
    return compile_mjcf(PENDULUM_XML, meshes={"bob": tris})


def box_model(half=0.1):
    from uhc_amd.model.mjcf import compile_mjcf
    return compile_mjcf(BOX_ON_PLANE_XML, meshes={"box": box_triangles(half, half, half)})
