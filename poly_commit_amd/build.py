"""Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU).

The library is split into translation units that compile in parallel (one per curve, one per
scalar field, the ABI glue, the hash-only kernels); objects are cached under csrc/_obj and
rebuilt when any source or header is newer."""
import concurrent.futures
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["abi.hip", "group.hip", "hash_tu.hip",
           "curve_bls12_381.hip", "curve_bn254.hip", "curve_pallas.hip",
           "field_bls12_381.hip", "field_bn254.hip", "field_pallas.hip"]
# PC_HIP_VARIANT=name builds an alternative library libpc_hip_<name>.so (own object cache) from the same sources with
# PC_HIP_CXXFLAGS -- kernel tuning experiments; load it with PC_HIP_LIB=<path>
_VARIANT = os.environ.get("PC_HIP_VARIANT", "")
OBJ = os.path.join(CSRC, "_obj" + ("_" + _VARIANT if _VARIANT else ""))
OUT = os.path.join(HERE, "libpc_hip" + ("_" + _VARIANT if _VARIANT else "") + ".so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def _newest_header():
    t = os.path.getmtime(os.path.join(os.path.dirname(HERE), "include", "pc_hip.h"))
    for f in os.listdir(CSRC):
        if f.endswith((".hpp", ".h")):
            t = max(t, os.path.getmtime(os.path.join(CSRC, f)))
    return t


def _stale_objects():
    th = _newest_header()
    flags_tag = " ".join(FLAGS + os.environ.get("PC_HIP_CXXFLAGS", "").split())
    tag_file = os.path.join(OBJ, "flags.txt")
    same_flags = os.path.exists(tag_file) and open(tag_file).read() == flags_tag
    stale = []
    for s in SOURCES:
        o = os.path.join(OBJ, s.replace(".hip", ".o"))
        src_t = max(th, os.path.getmtime(os.path.join(CSRC, s)))
        if not same_flags or not os.path.exists(o) or os.path.getmtime(o) < src_t:
            stale.append(s)
    return stale, flags_tag


def needs_build():
    if not os.path.exists(OUT):
        return True
    stale, _ = _stale_objects()
    if stale:
        return True
    return any(os.path.getmtime(os.path.join(OBJ, s.replace(".hip", ".o"))) > os.path.getmtime(OUT) for s in SOURCES)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    stale, flags_tag = _stale_objects()
    if force:
        stale = list(SOURCES)
    extra = os.environ.get("PC_HIP_CXXFLAGS", "").split()

    def compile_one(s):
        o = os.path.join(OBJ, s.replace(".hip", ".o"))
        cmd = ["hipcc"] + FLAGS + extra + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return o

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(stale) or 1, os.cpu_count() or 1)) as ex:
        list(ex.map(compile_one, stale))
    with open(os.path.join(OBJ, "flags.txt"), "w") as f:
        f.write(flags_tag)
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    import sys
    build(force="--force" in sys.argv, verbose=True)
