"""Builds libjmac_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose: bool = False, jobs: int = 4) -> str:
    cmd = ["make", "-C", os.path.join(HERE, "csrc"), "-j%d" % jobs]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libjmac_hip.so failed")
    return os.path.join(HERE, "libjmac_hip.so")


if __name__ == "__main__":
    print(build(verbose=True))
