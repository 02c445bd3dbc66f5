"""Build the HIP extension (librdc_assembly.so) in-tree for gfx950.

    python -m rdcfes_amd.build [--force] [-j N]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with gpurun snapshots.
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
OBJ = ROOT / "build"
LIBDIR = ROOT / "lib"
LIB = LIBDIR / "librdc_assembly.so"

SOURCES = [
    "rdc_capi.hip",
    "rdc_meshprep.cpp",
    "rdc_model_pihna.hip",
    "rdc_model_ripf.hip",
    "rdc_model_hcc.hip",
    "rdc_model_adpm.hip",
    "rdc_model_proteas.hip",
    "rdc_tet4_fast.hip",
    "rdc_tet4_ev.hip",
    "rdc_prep_ev.cpp",
    "rdc_solid.hip",
]

ARCH = os.environ.get("RDC_OFFLOAD_ARCH", "gfx950")
CXXFLAGS = [
    "-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-munsafe-fp-atomics",
    "-fopenmp", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas", "-Wno-pass-failed",
    "-ffp-contract=fast",
]
# experiments only: extra compiler flags for every HIP source, e.g. RDC_EXTRA_HIPCC_FLAGS="-mllvm -amdgpu-enable-max-ilp-scheduling-strategy=1"
CXXFLAGS += os.environ.get("RDC_EXTRA_HIPCC_FLAGS", "").split()


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built (there is no CPU fallback)")


def source_hash() -> str:
    """sha256 over the kernel sources: ties a committed profile (profiles/pmc_traffic.json) to the code it measured"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(CSRC.glob("*")):
        if f.suffix in (".hip", ".h", ".cpp"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def _newest_dep() -> float:
    deps = list(CSRC.glob("*")) + [ROOT.parent / "include" / "rdc_assembly.h", Path(__file__)]
    return max(p.stat().st_mtime for p in deps)


def _compile(src: str, extra=()) -> Path:
    OBJ.mkdir(exist_ok=True)
    obj = OBJ / (src.rsplit(".", 1)[0] + ".o")
    cmd = [hipcc(), *CXXFLAGS, *extra, "-x", "hip", "-c", str(CSRC / src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, jobs: int = 4, verbose: bool = True) -> Path:
    srcs = [s for s in SOURCES if (CSRC / s).exists()]
    if LIB.exists() and not force and LIB.stat().st_mtime >= _newest_dep():
        return LIB
    LIBDIR.mkdir(exist_ok=True)
    if verbose:
        print(f"[rdcfes_amd.build] compiling {len(srcs)} sources for {ARCH} ...", flush=True)
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(_compile, srcs))
    cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fopenmp", "-o", str(LIB), *map(str, objs)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[rdcfes_amd.build] built {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-j", type=int, default=4)
    a = ap.parse_args()
    build(force=a.force, jobs=a.j)
