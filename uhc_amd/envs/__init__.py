from .humanoid_im import HumanoidEnv, VecHumanoidEnv

env_dict = {"humanoid_im": HumanoidEnv}  # same key as uhc/envs/__init__.py:4-7 (the kinematic-policy env is out of scope)
