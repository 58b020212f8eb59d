"""Build-container-only helper: import the (pure-Python) reference from /root/reference with stub
modules standing in for the third-party packages this image lacks (mujoco_py, gym, smplx, ...).
Used by tools/gen_golden.py to produce input/output vectors and by tests/test_reference_live.py (skipped wherever
/root/reference is absent); bench.py, the GPU tests and the package never import this."""
import importlib
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


def assert_is_reference(obj):
    """Fail unless `obj` (module, class or function) was loaded from a file under REF.  This repository ships an alias package that is
    also called `uhc` (uhc/__init__.py -> uhc_amd); once it is in sys.modules a plain `from uhc... import X` silently yields this
    build's own X and a "comparison with the reference" compares the build with itself."""
    import inspect
    mod = obj if isinstance(obj, types.ModuleType) else sys.modules[obj.__module__]
    f = os.path.realpath(getattr(mod, "__file__", None) or inspect.getsourcefile(obj) or "")
    root = os.path.realpath(REF) + os.sep
    if not f.startswith(root):
        raise AssertionError(f"{getattr(obj, '__name__', obj)!r} comes from {f!r}, not from the reference under {root!r}")
    return obj


class reference_modules:
    """Context manager: inside it `import uhc...` resolves to the REFERENCE package, whatever was imported before.  Every `uhc` /
    `uhc.*` entry of sys.modules (this build's alias modules, if a test imported them earlier) is set aside, the alias finder is
    taken off sys.meta_path and REF goes to the front of sys.path; on exit the reference's modules are removed from sys.modules again
    and the previous state is restored, so later tests that use the alias are not handed reference code either.  Objects imported
    inside the block stay valid (they keep their modules alive)."""

    def __enter__(self):
        install()
        self._saved = {k: v for k, v in sys.modules.items() if k == "uhc" or k.startswith("uhc.")}
        for k in self._saved:
            del sys.modules[k]
        self._finders = [f for f in sys.meta_path if type(f).__name__ == "_AliasFinder"]
        for f in self._finders:
            sys.meta_path.remove(f)
        self._path = list(sys.path)
        sys.path[:] = [REF] + [p for p in sys.path if os.path.realpath(p or ".") != os.path.realpath(os.path.join(os.path.dirname(__file__), ".."))
                               and p != REF]
        importlib.invalidate_caches()
        import uhc
        assert_is_reference(uhc)
        return self

    def __exit__(self, *exc):
        for k in [k for k in sys.modules if k == "uhc" or k.startswith("uhc.")]:
            del sys.modules[k]
        sys.modules.update(self._saved)
        sys.path[:] = self._path
        for f in self._finders:
            if f not in sys.meta_path:
                sys.meta_path.insert(0, f)
        importlib.invalidate_caches()
        return False
