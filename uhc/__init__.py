"""`uhc` -- the reference's import path, served by this build.

The reference's callers write ``from uhc.agents import agent_dict``, ``from uhc.envs import env_dict``,
``from uhc.utils.config_utils.copycat_config import Config``, ``from uhc.data_loaders.dataset_amass_single import
DatasetAMASSSingle`` ... (scripts/train_uhc.py:30-32,90, scripts/eval_uhc.py).  Every ``uhc.X`` resolves to the module
``uhc_amd.X`` -- the same module object, not a copy -- so those scripts need no import rewrite.  Nothing is implemented here."""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import sys

import uhc_amd

__path__ = []  # a package without files of its own: sub-imports go through the finder below
_PREFIX, _REAL = "uhc.", "uhc_amd."


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if not name.startswith(_PREFIX):
            return None
        try:
            real = importlib.util.find_spec(_REAL + name[len(_PREFIX):])
        except ModuleNotFoundError:
            return None
        if real is None:
            return None
        return importlib.machinery.ModuleSpec(name, self, is_package=real.submodule_search_locations is not None)

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(_PREFIX):])

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())


def __getattr__(name):  # `import uhc; uhc.agents`
    try:
        return importlib.import_module(_PREFIX + name)
    except ModuleNotFoundError as e:
        raise AttributeError(name) from e
