"""ctypes mirrors of the POD parameter structs in include/rdc_assembly.h, plus constructors from the
reference's string-keyed parameter names (the keys ``es.parameters.get<Real>(...)`` reads:
src/pihna.C:358-381, src/ripf.C:377-408, src/coupled_hcc.C:450-461)."""
from __future__ import annotations

import ctypes as C

_D = C.c_double


class PihnaParams(C.Structure):
    _fields_ = [(n, _D) for n in (
        "time_step", "cells_min_capacity", "cells_max_capacity", "cytokines_max_capacity",
        "cells_max_capacity_exponent", "necrosis_c", "necrosis_h", "necrosis_v", "diffuse_c", "taxis_c",
        "diffuse_h", "taxis_h", "produce_c", "switch_c2h", "switch_h2c", "switch_h2n", "diffuse_v",
        "taxis_v", "produce_v", "secrete_a_c", "secrete_a_h", "uptake_a_v", "decay_a")]


class RipfParams(C.Structure):
    _fields_ = [(n, _D) for n in (
        "time_step", "VolFr_stroma", "VolFr_parenchyma", "VolFr_exponent", "VolFr_min_vacant",
        "VolFr_max_vacant", "phi_cc_B", "phi_cc_D", "phi_cc", "phi_fb_B", "phi_fb_D", "phi_fb", "phi_tol",
        "kappa", "kappa_RT_c", "delta", "delta_RT_a", "delta_RT_b", "lambda_", "lambda_RT_r", "lambda_HU_r",
        "omicro", "omicro_RT_r", "omicro_fb_b", "omega", "diffusion", "haptotaxis", "radiotaxis")] + [
        ("RT_dose_total_max", C.c_int32), ("_pad", C.c_int32)]


class HccParams(C.Structure):
    _fields_ = [(n, _D) for n in (
        "time_step", "cells_min_capacity", "cells_max_capacity", "cells_max_capacity_exponent", "produce_l",
        "diffuse_c", "mechano_c", "produce_c", "necrosis_l", "necrosis_c", "necrosis_pressure")]


class SolidMaterial(C.Structure):
    _fields_ = [("Young", _D), ("Poisson", _D), ("FibreStiffness", _D), ("rate", _D * 3)]


class SolidParams(C.Structure):
    _fields_ = [("pseudo_time", _D), ("displacement_penalty", _D), ("use_symmetry", C.c_int32),
                ("_pad", C.c_int32)]


class AdpmParams(C.Structure):
    """rdc_adpm_params, src/adpm.C:365-413."""
    _fields_ = [("time_step", _D), ("time", _D), ("decay_PrP_time_exponent", _D), ("decay_PrP", _D * 3),
                ("transform_A_b", _D * 5), ("transform_Tau", _D * 5), ("diffuse_A_b", _D * 3), ("taxis1_A_b", _D * 3),
                ("taxis2_A_b", _D * 3), ("produce_A_b", _D * 3), ("decay_A_b", _D * 3), ("diffuse_Tau", _D * 3),
                ("taxis1_Tau", _D * 3), ("taxis2_Tau", _D * 3), ("produce_Tau", _D * 3), ("decay_Tau", _D * 3),
                ("taxis_A_b_angle", _D), ("taxis_Tau_angle", _D)]


_PROTEAS_FIELDS = ["time_step", "cells_total_capacity", "RT_max_dosage", "host_proliferation", "host_vsc_threshold",
                   "host_RT_death_rate", "host_RT_exp_a", "host_RT_exp_b", "host_necrosis_rate", "tumour_diffusion",
                   "tumour_diffusion_host", "tumour_proliferation", "tumour_vsc_threshold", "tumour_RT_death_rate",
                   "tumour_RT_exp_a", "tumour_RT_exp_b", "tumour_necrosis_rate", "necrosis_clearance", "necrosis_slope",
                   "necrosis_vsc_threshold", "vascular_proliferation", "vascular_necrosis_rate", "oedema_diffusion",
                   "oedema_proliferation", "oedema_vsc_threshold", "oedema_RT_coeff", "oedema_RT_exp",
                   "oedema_reabsorption_rate"]


class ProteasParams(C.Structure):
    """rdc_proteas_params, src/proteas.C:376-409."""
    _fields_ = [(f, _D) for f in _PROTEAS_FIELDS]


class PihnaRanges(C.Structure):
    """rdc_pihna_ranges: the "range/*" keys of save_solution, src/pihna.C:853-861."""
    _fields_ = [(f, _D) for f in ("active_tumor_min", "active_tumor_max", "necrotic_min", "necrotic_max", "vascularity_min",
                                  "vascularity_max", "total_cell_min", "total_cell_max", "cells_max_capacity")]


class RipfRanges(C.Structure):
    """rdc_ripf_ranges: the "range_cc/*", "range_fb/*" keys of save_solution, src/ripf.C:790-795."""
    _fields_ = [(f, _D) for f in ("cc_HU_min", "cc_HU_max", "cc_min", "fb_HU_min", "fb_HU_max", "fb_min")]


class AdpmRanges(C.Structure):
    """rdc_adpm_ranges: the "range/A_b/*", "range/Tau/*" keys of save_solution, src/adpm.C:702-705."""
    _fields_ = [(f, _D) for f in ("A_b_min", "A_b_max", "Tau_min", "Tau_max")]


class RipfCheckParams(C.Structure):
    """rdc_ripf_check_params: what check_solution reads, src/ripf.C:697-703."""
    _fields_ = [("time_step", _D), ("HU_min", _D), ("HU_max", _D), ("RT_broad_fractions", C.c_int32),
                ("RT_focus_fractions", C.c_int32), ("day", C.c_int32), ("_pad", C.c_int32)]


# reference parameter key -> struct field
PIHNA_KEYS = {
    "time_step": "time_step",
    "cells_min_capacity": "cells_min_capacity",
    "cells_max_capacity": "cells_max_capacity",
    "cytokines_max_capacity": "cytokines_max_capacity",
    "cells_max_capacity/exponent": "cells_max_capacity_exponent",
    "necrosis/c": "necrosis_c", "necrosis/h": "necrosis_h", "necrosis/v": "necrosis_v",
    "diffuse/c": "diffuse_c", "taxis/c": "taxis_c", "diffuse/h": "diffuse_h", "taxis/h": "taxis_h",
    "produce/c": "produce_c", "switch/c/to/h": "switch_c2h", "switch/h/to/c": "switch_h2c",
    "switch/h/to/n": "switch_h2n", "diffuse/v": "diffuse_v", "taxis/v": "taxis_v", "produce/v": "produce_v",
    "secrete/a/from/c": "secrete_a_c", "secrete/a/from/h": "secrete_a_h", "uptake/a/from/v": "uptake_a_v",
    "decay/a": "decay_a",
}
# defaults of input(), src/pihna.C:139-235
PIHNA_DEFAULTS = {k: 0.0 for k in PIHNA_KEYS}
PIHNA_DEFAULTS.update({"time_step": 1.0e-9, "cells_min_capacity": 0.0, "cells_max_capacity": 1.0,
                       "cytokines_max_capacity": 1.0, "cells_max_capacity/exponent": 1.0})

RIPF_KEYS = {
    "time_step": "time_step",
    "volume_fraction/stroma": "VolFr_stroma", "volume_fraction/parenchyma": "VolFr_parenchyma",
    "volume_fraction/exponent": "VolFr_exponent", "volume_fraction/min_vacant": "VolFr_min_vacant",
    "volume_fraction/max_vacant": "VolFr_max_vacant",
    "HU/phi/cc/build": "phi_cc_B", "HU/phi/cc/decay": "phi_cc_D", "HU/phi/cc/rate": "phi_cc",
    "HU/phi/fb/build": "phi_fb_B", "HU/phi/fb/decay": "phi_fb_D", "HU/phi/fb/rate": "phi_fb",
    "HU/phi/tolerance": "phi_tol",
    "cc/kappa": "kappa", "cc/kappa/RT/c": "kappa_RT_c", "cc/delta": "delta", "cc/delta/RT/a": "delta_RT_a",
    "cc/delta/RT/b": "delta_RT_b",
    "fb/lambda": "lambda_", "fb/lambda/RT/r": "lambda_RT_r", "fb/lambda/HU/r": "lambda_HU_r",
    "fb/omicro": "omicro", "fb/omicro/RT/r": "omicro_RT_r", "fb/omicro/fb/b": "omicro_fb_b",
    "fb/omega": "omega", "fb/diffusion": "diffusion", "fb/haptotaxis": "haptotaxis",
    "fb/radiotaxis": "radiotaxis", "RT_dose/total/max": "RT_dose_total_max",
}
# defaults of input(), src/ripf.C:172-249
RIPF_DEFAULTS = {k: 0.0 for k in RIPF_KEYS}
RIPF_DEFAULTS.update({"time_step": 1.0e-9, "volume_fraction/exponent": 1.0, "volume_fraction/min_vacant": 1.0e-12,
                      "fb/lambda/HU/r": -1.0, "RT_dose/total/max": 0})

HCC_KEYS = {
    "time_step": "time_step", "cells/min_capacity": "cells_min_capacity", "cells/max_capacity": "cells_max_capacity",
    "cells/max_capacity/exponent": "cells_max_capacity_exponent", "produce/l": "produce_l",
    "diffuse/c": "diffuse_c", "mechano/c": "mechano_c", "produce/c": "produce_c", "necrosis/l": "necrosis_l",
    "necrosis/c": "necrosis_c", "necrosis/pressure": "necrosis_pressure",
}
HCC_DEFAULTS = {k: 0.0 for k in HCC_KEYS}
HCC_DEFAULTS.update({"time_step": 1.0e-9, "cells/max_capacity": 1.0, "cells/max_capacity/exponent": 1.0})


def _from_dict(cls, keys, defaults, d):
    unknown = set(d) - set(keys)
    if unknown:
        raise KeyError(f"unknown parameter key(s) for {cls.__name__}: {sorted(unknown)}")
    p = cls()
    for k, f in keys.items():
        v = d.get(k, defaults[k])
        setattr(p, f, int(v) if f == "RT_dose_total_max" else float(v))
    return p


def pihna_params_from_dict(d):
    return _from_dict(PihnaParams, PIHNA_KEYS, PIHNA_DEFAULTS, d)


def ripf_params_from_dict(d):
    p = _from_dict(RipfParams, RIPF_KEYS, RIPF_DEFAULTS, d)
    if "volume_fraction/max_vacant" not in d:  # src/ripf.C:180-182: default 1 - min_vacant
        p.VolFr_max_vacant = 1.0 - p.VolFr_min_vacant
    return p


def hcc_params_from_dict(d):
    return _from_dict(HccParams, HCC_KEYS, HCC_DEFAULTS, d)


# ADPM: reference key -> (struct field, index); defaults of input(), src/adpm.C:163-233
def _adpm_keys():
    keys = {"time_step": ("time_step", None), "decay/PrP/time_exponent": ("decay_PrP_time_exponent", None),
            "taxis/A_b/angle": ("taxis_A_b_angle", None), "taxis/Tau/angle": ("taxis_Tau_angle", None)}
    defaults = {"time_step": 1.0e-9, "decay/PrP/time_exponent": 0.0, "taxis/A_b/angle": 89.9, "taxis/Tau/angle": 89.9}
    def triple(key, field, kind, d0, d1):
        keys[key] = (field, 0); defaults[key] = 0.0
        keys[f"{key}/{kind}/0"] = (field, 1); defaults[f"{key}/{kind}/0"] = d0
        keys[f"{key}/{kind}/1"] = (field, 2); defaults[f"{key}/{kind}/1"] = d1
    triple("decay/PrP", "decay_PrP", "pulse", -1.0e-20, +1.0e+20)
    for sp in ("A_b", "Tau"):
        keys[f"transform/{sp}"] = (f"transform_{sp}", 0); defaults[f"transform/{sp}"] = 0.0
        for i, d in enumerate((-1.1e-20, -1.0e-20, +1.0e+20, +1.1e+20)):
            keys[f"transform/{sp}/trapezoid/{i}"] = (f"transform_{sp}", i + 1); defaults[f"transform/{sp}/trapezoid/{i}"] = d
        triple(f"diffuse/{sp}", f"diffuse_{sp}", "pulse", -1.0e-20, +1.0e+20)
        triple(f"taxis_1/{sp}", f"taxis1_{sp}", "pulse", -1.0e-20, +1.0e+20)
        triple(f"taxis_2/{sp}", f"taxis2_{sp}", "pulse", -1.0e-20, +1.0e+20)
        triple(f"produce/{sp}", f"produce_{sp}", "sigmoid", +1.0e+20, +1.1e+20)
        triple(f"decay/{sp}", f"decay_{sp}", "pulse", -1.0e-20, +1.0e+20)
    return keys, defaults


ADPM_KEYS, ADPM_DEFAULTS = _adpm_keys()


def adpm_params_from_dict(d, time=0.0):
    """Reference-keyed dict -> AdpmParams.  The two angles are given in DEGREES as in the input file
    (input() converts them, src/adpm.C:193,215); `time` is system.time of the step being assembled."""
    import math
    unknown = set(d) - set(ADPM_KEYS)
    if unknown:
        raise KeyError(f"unknown ADPM parameter keys: {sorted(unknown)}")
    p = AdpmParams()
    p.time = float(time)
    for key, (field, idx) in ADPM_KEYS.items():
        v = float(d.get(key, ADPM_DEFAULTS[key]))
        if key.endswith("/angle"):
            v = math.radians(v)
        if idx is None:
            setattr(p, field, v)
        else:
            getattr(p, field)[idx] = v
    return p


# PROTEAS: reference key -> struct field; defaults of input(), src/proteas.C:135,180-212
PROTEAS_KEYS = {"time_step": "time_step", "cells/total_capacity": "cells_total_capacity",
                "radiotherapy/max_dosage": "RT_max_dosage"}
for _f in _PROTEAS_FIELDS[3:]:
    _grp, _rest = _f.split("_", 1)
    PROTEAS_KEYS[f"{_grp}/{_rest}"] = _f
PROTEAS_DEFAULTS = {k: 1.0 for k in PROTEAS_KEYS}
PROTEAS_DEFAULTS["time_step"] = 1.0e-9


def proteas_params_from_dict(d):
    unknown = set(d) - set(PROTEAS_KEYS)
    if unknown:
        raise KeyError(f"unknown PROTEAS parameter keys: {sorted(unknown)}")
    p = ProteasParams()
    for key, field in PROTEAS_KEYS.items():
        setattr(p, field, float(d.get(key, PROTEAS_DEFAULTS[key])))
    return p
