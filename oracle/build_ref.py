"""TEST INFRASTRUCTURE ONLY -- builds the reference's own C++ persistence extension into oracle/_ref/.

Compiles, from the sources where they lie under /root/reference (never copied), the vendored
TopologyLayer persistence extension
(nnUNet/nnunetv2/training/topologylayer/functional/persistence/{cocycle,complex,hom,cohom,pybind}.cpp)
with g++ through torch.utils.cpp_extension (the reference ships no build script of its own for it).
Outputs go only to oracle/_ref/ (git-ignored AND gpurun-ignored since round 3: it stays in the build container; the GPU
box sees its outputs through tests/golden/persistence_grid.json).
Used by tests/ and tools/make_golden.py to pin oracle/cc_oracle.c; the product path never loads it.
"""
import os
import sys

REF = "/root/reference/nnUNet/nnunetv2/training/topologylayer/functional/persistence"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
NAME = "mvd_ref_persistence"


def built_path():
    for f in (os.listdir(OUT) if os.path.isdir(OUT) else []):
        if f.startswith(NAME) and f.endswith(".so"):
            return os.path.join(OUT, f)
    return None


def build(verbose=False):
    """Build (if the reference tree is present) and return the .so path, else the prebuilt one or None."""
    if not os.path.isdir(REF):
        return built_path()
    os.makedirs(OUT, exist_ok=True)
    from torch.utils import cpp_extension
    srcs = [os.path.join(REF, f) for f in ("cocycle.cpp", "complex.cpp", "hom.cpp", "cohom.cpp", "pybind.cpp")]
    cpp_extension.load(name=NAME, sources=srcs, build_directory=OUT, extra_cflags=["-O2", "-w"],
                       verbose=verbose, is_python_module=False)
    return built_path()


def load():
    """Import the prebuilt module from oracle/_ref (does not need /root/reference)."""
    p = built_path()
    if p is None:
        return None
    import importlib.util
    import torch  # noqa: F401  (libtorch symbols must be loaded first)
    spec = importlib.util.spec_from_file_location(NAME, p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
