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
    "rdc_tet4_evc.hip",
    "rdc_prep_ev.cpp",
    "rdc_solid.hip",
    "rdc_solid_cl.hip",
    "rdc_prep_cl.cpp",
]

ARCH = os.environ.get("RDC_OFFLOAD_ARCH", "gfx950")
CXXFLAGS = [
    "-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-munsafe-fp-atomics",
    "-fopenmp", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas", "-Wno-pass-failed",
    "-Wno-inline-asm",   # "clobber list contains reserved registers: m0": the LDS-DMA asm sets m0 itself, on purpose
    "-ffp-contract=fast",
]
# experiments only: extra compiler flags for every HIP source, e.g. RDC_EXTRA_HIPCC_FLAGS="-mllvm -amdgpu-enable-max-ilp-scheduling-strategy=1"
CXXFLAGS += os.environ.get("RDC_EXTRA_HIPCC_FLAGS", "").split()
RESOURCES = LIBDIR / "kernel_resources.json"   # per kernel: registers, scratch, occupancy (from -Rpass-analysis)


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


def _parse_resources(stderr: str) -> dict:
    """kernel-resource-usage remarks of one translation unit -> {mangled kernel name: {vgprs, agprs, scratch, ...}}"""
    import re
    out, cur = {}, None
    keys = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
            "Occupancy [waves/SIMD]": "waves_per_simd", "VGPRs Spill": "vgpr_spills", "SGPRs Spill": "sgpr_spills",
            "LDS Size [bytes/block]": "static_lds_bytes"}
    for ln in stderr.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", ln)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", ln)
        if m and cur is not None and m.group(1).strip() in keys:
            cur[keys[m.group(1).strip()]] = int(m.group(2))
    return out


def _compile(src: str, extra=()):
    OBJ.mkdir(exist_ok=True)
    obj = OBJ / (src.rsplit(".", 1)[0] + ".o")
    hip = src.endswith(".hip")
    cmd = [hipcc(), *CXXFLAGS, *extra, *(["-Rpass-analysis=kernel-resource-usage"] if hip else []), "-x", "hip", "-c",
           str(CSRC / src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    res = _parse_resources(r.stderr) if hip else {}
    # everything that is not a kernel-resource-usage remark (= genuine warnings, with their source excerpts) reaches the log
    import re
    keep, in_diag = [], False
    for ln in r.stderr.splitlines():
        if re.search(r": (warning|error|fatal error|note):", ln):
            in_diag = True
        elif re.search(r": remark:", ln) or re.match(r"\d+ (warnings?|remarks?|errors?)", ln.strip()):
            in_diag = False
        if in_diag and ln.strip():
            keep.append(ln)
    if keep:
        sys.stderr.write(f"[rdcfes_amd.build] {src}:\n" + "\n".join(keep) + "\n")
    return obj, res


def build(force: bool = False, jobs: int = 4, verbose: bool = True) -> Path:
    srcs = [s for s in SOURCES if (CSRC / s).exists()]
    if LIB.exists() and not force and LIB.stat().st_mtime >= _newest_dep():
        return LIB
    LIBDIR.mkdir(exist_ok=True)
    if verbose:
        print(f"[rdcfes_amd.build] compiling {len(srcs)} sources for {ARCH} ...", flush=True)
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        done = list(ex.map(_compile, srcs))
    objs = [d[0] for d in done]
    resources = {}
    for _, res in done:
        resources.update(res)
    cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fopenmp", "-o", str(LIB), *map(str, objs)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    import json
    RESOURCES.write_text(json.dumps(resources, indent=0, sort_keys=True))
    # spills / scratch are performance bugs on this path: always say so.  The kernels every default configuration runs
    # are listed by name; the experimental / diagnostic instantiations are counted (full list: lib/kernel_resources.json,
    # RDC_BUILD_VERBOSE=1).  RDC_BUILD_STRICT=1 turns scratch in a default-path kernel into an error.
    default_path = ("k_tet4_ev", "k_tet4_evm", "k_tet4_evc", "k_tet4_rg5", "k_hex8_cl", "k_solid_cl", "k_pack_nodes")
    bad = {k: v for k, v in resources.items() if v.get("scratch_bytes_per_lane", 0) or v.get("vgpr_spills", 0)}
    hot = {k: v for k, v in bad.items() if any(f"{len(n)}{n}I" in k or f"{len(n)}{n}E" in k for n in default_path)}
    for k, v in sorted(bad.items() if os.environ.get("RDC_BUILD_VERBOSE") else hot.items()):
        sys.stderr.write(f"[rdcfes_amd.build] scratch: {k}: {v.get('scratch_bytes_per_lane', 0)} B/lane, {v.get('vgpr_spills', 0)} VGPR spills, "
                         f"{v.get('vgprs')} VGPRs\n")
    if bad:
        sys.stderr.write(f"[rdcfes_amd.build] {len(bad)} of {len(resources)} kernel instantiations use scratch ({len(hot)} on a default path)\n")
    if hot and os.environ.get("RDC_BUILD_STRICT"):
        raise RuntimeError(f"{len(hot)} default-path kernels use scratch memory")
    if verbose:
        print(f"[rdcfes_amd.build] built {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-j", type=int, default=4)
    a = ap.parse_args()
    build(force=a.force, jobs=a.j)
