"""`uhc` -- the reference's import path, served by this build.

The reference's callers write ``from uhc.agents import agent_dict``, ``from uhc.envs import env_dict``,
``from uhc.utils.config_utils.copycat_config import Config``, ``from uhc.data_loaders.dataset_amass_single import
DatasetAMASSSingle`` ... (scripts/train_uhc.py:30-32,90, scripts/eval_uhc.py).  Every ``uhc.X`` resolves to the module
``uhc_amd.X`` -- the same module object, not a copy -- so those scripts need no import rewrite.  Nothing is implemented here.

Where the reference spreads one package over several files and this build keeps it in one module, the file-level paths resolve to
that module (`_SPLIT`): ``from uhc.khrylib.rl.core.policy_gaussian import PolicyGaussian``, ``...core.critic import Value``,
``uhc.khrylib.rl.agents.agent_ppo``, ``uhc.smpllib.smpl_parser`` (its constants live in smpl_mujoco) ... -- every ``from uhc...`` line
of uhc/agents/agent_copycat.py:31-48 and of the header of uhc/envs/humanoid_im.py:14-44 (tests/test_boundary_cpu.py)."""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import sys

import uhc_amd

__path__ = []  # a package without files of its own: sub-imports go through the finder below
_PREFIX, _REAL = "uhc.", "uhc_amd."
# reference module (file) -> the module of this build that defines the same names
_SPLIT = {
    **{"khrylib.rl.core." + m: "khrylib.rl.core" for m in ("policy_gaussian", "critic", "common", "trajbatch", "logger_rl", "policy", "distributions")},
    **{"khrylib.rl.agents." + m: "khrylib.rl.agents" for m in ("agent", "agent_pg", "agent_ppo")},
    "khrylib.utils.transformation": "utils.transformation",  # (the reference carries two copies of the module)
    "smpllib.smpl_parser": "smpllib.smpl_mujoco",              # bone-name tables (SMPL_BONE_ORDER_NAMES, SMPL_EE_NAMES, SMPLH_BONE_ORDER_NAMES ...)
}


def _real_name(name):
    rel = name[len(_PREFIX):]
    return _REAL + _SPLIT.get(rel, rel)


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if not name.startswith(_PREFIX):
            return None
        try:
            real = importlib.util.find_spec(_real_name(name))
        except ModuleNotFoundError:
            return None
        if real is None:
            return None
        return importlib.machinery.ModuleSpec(name, self, is_package=real.submodule_search_locations is not None)

    def create_module(self, spec):
        return importlib.import_module(_real_name(spec.name))

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())


def __getattr__(name):  # `import uhc; uhc.agents`
    try:
        return importlib.import_module(_PREFIX + name)
    except ModuleNotFoundError as e:
        raise AttributeError(name) from e
