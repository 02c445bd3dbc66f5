"""rdcfes_amd — MI355X-native element assembly for the rdcFEs reaction-diffusion-convection models.

The compute path is the HIP library ``lib/librdc_assembly.so`` (C-ABI in ``include/rdc_assembly.h``).
This Python package is the test / benchmark harness above that ABI plus the host-side partitioning
logic; it contains no numerical fallback: without the built library and a GPU every compute call
raises.
"""
from .params import (PihnaParams, RipfParams, HccParams, SolidParams, SolidMaterial, RipfCheckParams, AdpmParams, ProteasParams, PihnaRanges, RipfRanges, AdpmRanges,
                     pihna_params_from_dict, ripf_params_from_dict, hcc_params_from_dict,
                     adpm_params_from_dict, proteas_params_from_dict)
from .context import (AssemblyContext, RdcError, TET4, HEX8, SCATTER_AUTO, SCATTER_COLOURED,
                      SCATTER_ROWGATHER, FIELD_OLD_SOLUTION, FIELD_AUX_NODAL,
                      FIELD_UNDEFORMED_XYZ, FIELD_ELEM_FIBRE, FIELD_PREV_SOLUTION, FIELD_TIME_DERIV,
                      FIELD_RT_DOSE, FIELD_ELEM_TRACTS)

__all__ = [
    "PihnaParams", "RipfParams", "HccParams", "SolidParams", "SolidMaterial", "RipfCheckParams", "AdpmParams", "adpm_params_from_dict", "ProteasParams", "proteas_params_from_dict", "PihnaRanges", "RipfRanges", "AdpmRanges",
    "pihna_params_from_dict", "ripf_params_from_dict", "hcc_params_from_dict",
    "AssemblyContext", "RdcError", "TET4", "HEX8", "SCATTER_AUTO", "SCATTER_COLOURED",
    "SCATTER_ROWGATHER", "FIELD_OLD_SOLUTION", "FIELD_AUX_NODAL", "FIELD_UNDEFORMED_XYZ",
    "FIELD_ELEM_FIBRE", "FIELD_PREV_SOLUTION", "FIELD_TIME_DERIV", "FIELD_RT_DOSE", "FIELD_ELEM_TRACTS",
]
