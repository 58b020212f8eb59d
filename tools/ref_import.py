"""Build-container-only helper: import the (pure-Python) reference from /root/reference with stub
modules standing in for the third-party packages this image lacks (mujoco_py, gym, smplx, ...).
Used by tools/gen_golden.py to produce input/output vectors and by tests/test_reference_live.py (skipped wherever
/root/reference is absent); bench.py, the GPU tests and the package never import this."""
import importlib.abc
import importlib.machinery
import os
import sys
import types

REF = os.environ.get("UHC_REFERENCE", "/root/reference")
MISSING = ["mujoco_py", "mujoco", "gym", "lxml", "smplx", "cv2", "wandb", "stl", "vtk", "glfw", "fasteners", "ipdb",
           "imageio", "torchgeometry", "human_body_prior", "autograd", "chumpy", "OpenGL", "pyglet", "matplotlib",
           "mpl_toolkits", "skimage", "PIL", "open3d", "trimesh", "pyvista", "numpy_stl", "gdown", "termcolor", "tensorboardX",
           "seaborn", "imageio_ffmpeg", "mediapy", "pytorch3d", "numba", "sklearn_extra", "easydict", "pytorch_lightning",
           "torchvision", "scenepic", "tqdm_batch"]


class _Anything:
    """Attribute sink that can be called, subclassed, indexed and iterated (empty)."""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Anything()

    def __call__(self, *a, **k):
        return _Anything()

    def __getitem__(self, k):
        return _Anything()

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        if name[:1].isupper():
            return type(name, (object,), {"__init__": lambda self, *a, **k: None})
        return _Anything()


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        root = fullname.split(".")[0]
        if root in MISSING:
            try:
                # prefer a real installation if one exists
                for f in sys.meta_path:
                    if f is self:
                        continue
                    spec = f.find_spec(fullname, path, target) if hasattr(f, "find_spec") else None
                    if spec is not None:
                        return spec
            except Exception:
                pass
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def install():
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder())
    if REF not in sys.path:
        sys.path.insert(0, REF)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
