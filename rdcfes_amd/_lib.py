"""Loader of the HIP extension.  Fails loudly: there is no Python/CPU substitute for it."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

from .params import HccParams, PihnaParams, RipfParams, SolidMaterial, SolidParams

LIB_PATH = Path(__file__).resolve().parent / "lib" / "librdc_assembly.so"

_lib = None

i64, i32, u32, dbl = C.c_int64, C.c_int32, C.c_uint32, C.c_double
P = C.POINTER
ctx_p = C.c_void_p

# name -> (restype, argtypes): every symbol include/rdc_assembly.h declares
SIGNATURES = {
    "rdc_abi_version": (C.c_int, []),
    "rdc_device_count": (C.c_int, [P(C.c_int)]),
    "rdc_ctx_create": (C.c_int, [C.c_int, P(ctx_p)]),
    "rdc_ctx_destroy": (C.c_int, [ctx_p]),
    "rdc_last_error": (C.c_char_p, [ctx_p]),
    "rdc_set_stream": (C.c_int, [ctx_p, C.c_void_p]),
    "rdc_synchronize": (C.c_int, [ctx_p]),
    "rdc_set_scatter": (C.c_int, [ctx_p, C.c_int]),
    "rdc_get_scatter": (C.c_int, [ctx_p, P(C.c_int)]),
    "rdc_set_kernel_variant": (C.c_int, [ctx_p, C.c_int]),
    "rdc_set_option": (C.c_int, [ctx_p, C.c_char_p, C.c_int]),
    "rdc_mesh_upload": (C.c_int, [ctx_p, C.c_int, i64, i64, i64, P(u32), P(dbl), C.c_int]),
    "rdc_mesh_update_coords": (C.c_int, [ctx_p, P(dbl)]),
    "rdc_mesh_coords_device_ptr": (C.c_int, [ctx_p, P(C.c_void_p)]),
    "rdc_mesh_dims": (C.c_int, [ctx_p, P(i64), P(i64), P(i64), P(C.c_int), P(C.c_int), P(C.c_int)]),
    "rdc_csr_dims": (C.c_int, [ctx_p, P(i64), P(i64)]),
    "rdc_csr_pattern_download": (C.c_int, [ctx_p, P(i64), P(i32)]),
    "rdc_mesh_colours_download": (C.c_int, [ctx_p, P(i32)]),
    "rdc_field_upload": (C.c_int, [ctx_p, C.c_int, P(dbl), i64]),
    "rdc_field_download": (C.c_int, [ctx_p, C.c_int, P(dbl), i64]),
    "rdc_field_device_ptr": (C.c_int, [ctx_p, C.c_int, i64, P(C.c_void_p)]),
    "rdc_field_bind_device": (C.c_int, [ctx_p, C.c_int, C.c_void_p, i64]),
    "rdc_solid_set_materials": (C.c_int, [ctx_p, P(i32), i32, P(SolidMaterial)]),
    "rdc_solid_set_sides": (C.c_int, [ctx_p, i64, P(i64), P(i32), P(dbl)]),
    "rdc_assemble_pihna": (C.c_int, [ctx_p, P(PihnaParams)]),
    "rdc_assemble_ripf": (C.c_int, [ctx_p, P(RipfParams)]),
    "rdc_assemble_hcc": (C.c_int, [ctx_p, P(HccParams)]),
    "rdc_solid_assemble": (C.c_int, [ctx_p, P(SolidParams), C.c_int]),
    "rdc_assemble_pihna_part": (C.c_int, [ctx_p, P(PihnaParams), C.c_int, C.c_void_p]),
    "rdc_assemble_hcc_part": (C.c_int, [ctx_p, P(HccParams), C.c_int, C.c_void_p]),
    "rdc_solid_assemble_part": (C.c_int, [ctx_p, P(SolidParams), C.c_int, C.c_int, C.c_void_p]),
    "rdc_csr_values_device_ptr": (C.c_int, [ctx_p, P(C.c_void_p), P(C.c_void_p)]),
    "rdc_csr_download": (C.c_int, [ctx_p, P(dbl), P(dbl)]),
    "rdc_csr_download_rows": (C.c_int, [ctx_p, i64, i64, C.c_void_p, C.c_void_p, C.c_int]),
    "rdc_part1_nodes": (C.c_int, [ctx_p, P(i64)]),
    "rdc_csr_download_rows_async": (C.c_int, [ctx_p, i64, i64, C.c_void_p, C.c_void_p, P(C.c_int)]),
    "rdc_ticket_wait": (C.c_int, [ctx_p, C.c_int]),
    "rdc_host_pin": (C.c_int, [ctx_p, C.c_void_p, C.c_size_t]),
    "rdc_host_unpin": (C.c_int, [ctx_p, C.c_void_p]),
    "rdc_assemble_adpm": (C.c_int, [ctx_p, C.c_void_p]),
    "rdc_assemble_proteas": (C.c_int, [ctx_p, C.c_void_p]),
    "rdc_clamp_nonnegative": (C.c_int, [ctx_p, C.c_int]),
    "rdc_pihna_volume_integrals": (C.c_int, [ctx_p, C.c_void_p, i64, C.POINTER(C.c_double)]),
    "rdc_ripf_volume_integrals": (C.c_int, [ctx_p, C.c_void_p, i64, C.POINTER(C.c_double)]),
    "rdc_adpm_parcellation_integrals": (C.c_int, [ctx_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, i64,
                                                  C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "rdc_solid_post_process": (C.c_int, [ctx_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "rdc_ripf_check_solution": (C.c_int, [ctx_p, C.c_void_p, C.POINTER(C.c_double)]),
    "rdc_timing_enable": (C.c_int, [ctx_p, C.c_int]),
    "rdc_timing_last_ms": (C.c_int, [ctx_p, P(C.c_float)]),
    "rdc_timing_sum_ms": (C.c_int, [ctx_p, P(C.c_float), P(C.c_int)]),
    "rdc_timing_samples_ms": (C.c_int, [ctx_p, P(C.c_float), C.c_int, P(C.c_int)]),
    "rdc_debug_stamps": (C.c_int, [ctx_p, P(C.c_longlong), i64, P(i64)]),
}


def header_abi_version() -> int:
    import re
    h = (Path(__file__).resolve().parent.parent / "include" / "rdc_assembly.h").read_text()
    return int(re.search(r"#define\s+RDC_ABI_VERSION\s+(\d+)", h).group(1))


def load():
    """Load librdc_assembly.so and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m rdcfes_amd.build` "
            "(the assembly path is HIP-only; there is no CPU fallback)")
    lib = C.CDLL(str(LIB_PATH))
    # version first: an older .so then fails with a version message, not with a missing-symbol AttributeError
    lib.rdc_abi_version.restype = C.c_int
    want, have = header_abi_version(), lib.rdc_abi_version()
    if have != want:
        raise RuntimeError(f"{LIB_PATH} has ABI version {have}, include/rdc_assembly.h declares {want}: rebuild it "
                           "(`python -m rdcfes_amd.build --force`)")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
