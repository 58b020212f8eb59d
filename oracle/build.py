"""Build recipe for the oracle's C restatement (test infrastructure).  gcc only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libphysics_oracle.so")


def build(force: bool = False) -> str:
    src = [os.path.join(HERE, "physics_oracle.c")]
    deps = src + [os.path.join(HERE, "physics_oracle.h"), os.path.join(HERE, "..", "include", "uhc_amd.h")]
    if not force and os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(p) for p in deps):
        return SO
    cmd = ["gcc", "-O3", "-march=x86-64-v2", "-std=gnu99", "-fPIC", "-shared", "-fopenmp", "-Wall", "-o", SO] + src + ["-lm"]
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force=True))
