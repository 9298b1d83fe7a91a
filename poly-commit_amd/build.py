"""Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["csrc/pc_hip.hip"]
OUT = os.path.join(HERE, "libpc_hip.so")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    for root, _, files in os.walk(os.path.join(HERE, "csrc")):
        for f in files:
            if os.path.getmtime(os.path.join(root, f)) > t:
                return True
    inc = os.path.join(os.path.dirname(HERE), "include", "pc_hip.h")
    return os.path.getmtime(inc) > t


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-value"] + os.environ.get("PC_HIP_CXXFLAGS", "").split() + \
          ["-o", OUT] + [os.path.join(HERE, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force=True, verbose=True)
