"""Thin object wrapper over the C-ABI (include/rdc_assembly.h); one instance == one rdc_ctx."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .params import HccParams, PihnaParams, RipfParams, SolidMaterial, SolidParams

TET4, HEX8 = 4, 8
SCATTER_AUTO, SCATTER_COLOURED, SCATTER_ROWGATHER = 0, 1, 2
FIELD_OLD_SOLUTION, FIELD_AUX_NODAL, FIELD_UNDEFORMED_XYZ, FIELD_ELEM_FIBRE = 0, 1, 2, 3
FIELD_PREV_SOLUTION, FIELD_TIME_DERIV, FIELD_RT_DOSE = 4, 5, 6
FIELD_ELEM_TRACTS = FIELD_ELEM_FIBRE  # ADPM: same per-element slot
VARIANT_AUTO, VARIANT_GENERIC = 0, 1


class RdcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rdc error {code}: {msg}")
        self.code = code


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class AssemblyContext:
    """Owns one rdc_ctx (one GPU, one mesh partition, one system)."""

    def __init__(self, device: int = 0):
        self._lib = _lib.load()
        h = C.c_void_p()
        rc = self._lib.rdc_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise RdcError(rc, self._lib.rdc_last_error(None).decode())
        self._h = h
        self.device = int(device)
        self.n_elem = self.n_node = self.n_owned = 0
        self.nvar = 0

    # -- plumbing
    def _ck(self, rc):
        if rc != 0:
            raise RdcError(rc, self._lib.rdc_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rdc_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_stream(self, hip_stream_ptr):
        self._ck(self._lib.rdc_set_stream(self._h, C.c_void_p(int(hip_stream_ptr) if hip_stream_ptr else None)))

    def synchronize(self):
        self._ck(self._lib.rdc_synchronize(self._h))

    def set_scatter(self, strategy):
        self._ck(self._lib.rdc_set_scatter(self._h, int(strategy)))

    def set_kernel_variant(self, variant):
        self._ck(self._lib.rdc_set_kernel_variant(self._h, int(variant)))

    def set_option(self, key, value):
        self._ck(self._lib.rdc_set_option(self._h, key.encode(), int(value)))

    def get_scatter(self):
        s = C.c_int()
        self._ck(self._lib.rdc_get_scatter(self._h, C.byref(s)))
        return s.value

    # -- mesh
    def mesh_upload(self, elem_type, conn, xyz, nvar, n_owned=None):
        conn = np.ascontiguousarray(conn, dtype=np.uint32)
        xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        if conn.ndim != 2 or conn.shape[1] != elem_type:
            raise ValueError("conn must be [n_elem][elem_type]")
        if xyz.ndim != 2 or xyz.shape[1] != 3:
            raise ValueError("xyz must be [n_node][3]")
        n_owned = xyz.shape[0] if n_owned is None else int(n_owned)
        self._ck(self._lib.rdc_mesh_upload(self._h, int(elem_type), conn.shape[0], xyz.shape[0], n_owned,
                                           conn.ctypes.data_as(C.POINTER(C.c_uint32)), _dp(xyz), int(nvar)))
        self.elem_type, self.n_elem, self.n_node, self.n_owned, self.nvar = elem_type, conn.shape[0], xyz.shape[0], n_owned, nvar

    def mesh_update_coords(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        if xyz.shape != (self.n_node, 3):
            raise ValueError("xyz shape mismatch")
        self._ck(self._lib.rdc_mesh_update_coords(self._h, _dp(xyz)))

    def coords_device_ptr(self):
        """device address of the CURRENT node coordinates [n_node][3] (rdc_mesh_coords_device_ptr)"""
        p = C.c_void_p()
        self._ck(self._lib.rdc_mesh_coords_device_ptr(self._h, C.byref(p)))
        return p.value

    def coords_tensor(self):
        """zero-copy torch view [n_node][3] of the context's current coordinates: the moving-mesh models update them in
        place on the device (mesh = solution of the solid system) and the halo exchange writes their ghost rows"""
        import torch

        class _View:
            pass
        v = _View()
        v.__cuda_array_interface__ = {"shape": (self.n_node, 3), "typestr": "<f8", "data": (self.coords_device_ptr(), False),
                                      "version": 2, "strides": None}
        return torch.as_tensor(v, device=torch.device("cuda", self.device))

    def n_colours(self):
        nc = C.c_int()
        self._ck(self._lib.rdc_mesh_dims(self._h, None, None, None, None, None, C.byref(nc)))
        return nc.value

    def colours(self):
        out = np.empty(self.n_elem, dtype=np.int32)
        self._ck(self._lib.rdc_mesh_colours_download(self._h, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def csr_dims(self):
        r, z = C.c_int64(), C.c_int64()
        self._ck(self._lib.rdc_csr_dims(self._h, C.byref(r), C.byref(z)))
        return r.value, z.value

    def csr_pattern(self):
        n_rows, nnz = self.csr_dims()
        row_ptr = np.empty(n_rows + 1, dtype=np.int64)
        col = np.empty(nnz, dtype=np.int32)
        self._ck(self._lib.rdc_csr_pattern_download(self._h, row_ptr.ctypes.data_as(C.POINTER(C.c_int64)),
                                                    col.ctypes.data_as(C.POINTER(C.c_int32))))
        return row_ptr, col

    # -- fields
    def field_upload(self, field, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        self._ck(self._lib.rdc_field_upload(self._h, int(field), _dp(arr), arr.size))

    def field_download(self, field, count):
        out = np.empty(int(count), dtype=np.float64)
        self._ck(self._lib.rdc_field_download(self._h, int(field), _dp(out), out.size))
        return out

    def field_device_ptr(self, field, count):
        p = C.c_void_p()
        self._ck(self._lib.rdc_field_device_ptr(self._h, int(field), int(count), C.byref(p)))
        return p.value

    def field_bind_device(self, field, dptr, count):
        self._ck(self._lib.rdc_field_bind_device(self._h, int(field), C.c_void_p(int(dptr)), int(count)))

    def clamp_nonnegative(self, field):
        self._ck(self._lib.rdc_clamp_nonnegative(self._h, int(field)))

    def pihna_volume_integrals(self, ranges, n_elem=-1):
        """the four volume sums of PIHNA's CSV line (src/pihna.C:842-976) over the first n_elem elements"""
        out = np.zeros(4)
        self._ck(self._lib.rdc_pihna_volume_integrals(self._h, C.byref(ranges), int(n_elem), _dp(out)))
        return out

    def ripf_volume_integrals(self, ranges, n_elem=-1):
        """Tumour_Volume, Fibrosis_Volume of RIPF's CSV line (src/ripf.C:777-866) over the first n_elem elements"""
        out = np.zeros(2)
        self._ck(self._lib.rdc_ripf_volume_integrals(self._h, C.byref(ranges), int(n_elem), _dp(out)))
        return out

    def adpm_parcellation_integrals(self, ranges, elem_subdomain, ids, n_elem=-1):
        """ADPM's CSV line (src/adpm.C:690-829): -> ([n_ids][4] = A_b concentration, Tau concentration, A_b volume,
        Tau volume per parcellation id; last element of every region, -1 if it has none here)"""
        sub = np.ascontiguousarray(elem_subdomain, dtype=np.int32)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        if sub.size < (self.n_elem if n_elem < 0 else n_elem):
            raise ValueError("elem_subdomain shorter than the element range")
        out, last = np.zeros((ids.size, 4)), np.full(ids.size, -1, dtype=np.int64)
        self._ck(self._lib.rdc_adpm_parcellation_integrals(
            self._h, C.byref(ranges), sub.ctypes.data_as(C.POINTER(C.c_int32)), ids.ctypes.data_as(C.POINTER(C.c_int32)),
            int(ids.size), int(n_elem), _dp(out), last.ctypes.data_as(C.POINTER(C.c_int64))))
        return out, last

    def solid_post_process(self, params):
        """SolidSystem::post_process -> (pressure [ne], von_mises [ne], fibre_current [ne][3])"""
        pr, vm, fc = np.empty(self.n_elem), np.empty(self.n_elem), np.empty((self.n_elem, 3))
        self._ck(self._lib.rdc_solid_post_process(self._h, C.byref(params), _dp(pr), _dp(vm), _dp(fc)))
        return pr, vm, fc

    def ripf_check_solution(self, params):
        """RIPF check_solution on the device fields (src/ripf.C:675-775); returns the maximum total RT dose."""
        mx = C.c_double(0.0)
        self._ck(self._lib.rdc_ripf_check_solution(self._h, C.byref(params), C.byref(mx)))
        return mx.value

    # -- solid set-up
    def solid_set_materials(self, elem_material, materials):
        em = np.ascontiguousarray(elem_material, dtype=np.int32)
        if em.shape != (self.n_elem,):
            raise ValueError("elem_material must have one entry per element")
        arr = (SolidMaterial * len(materials))(*materials)
        self._ck(self._lib.rdc_solid_set_materials(self._h, em.ctypes.data_as(C.POINTER(C.c_int32)), len(materials), arr))

    def solid_set_sides(self, side_elem, side_id, side_disp):
        se = np.ascontiguousarray(side_elem, dtype=np.int64)
        si = np.ascontiguousarray(side_id, dtype=np.int32)
        sd = np.ascontiguousarray(side_disp, dtype=np.float64).reshape(-1, 3)
        if not (se.shape[0] == si.shape[0] == sd.shape[0]):
            raise ValueError("side arrays must have equal length")
        self._ck(self._lib.rdc_solid_set_sides(self._h, se.shape[0], se.ctypes.data_as(C.POINTER(C.c_int64)),
                                               si.ctypes.data_as(C.POINTER(C.c_int32)), _dp(sd)))

    # -- the hot path
    def assemble_pihna(self, p: PihnaParams):
        self._ck(self._lib.rdc_assemble_pihna(self._h, C.byref(p)))

    def assemble_ripf(self, p: RipfParams):
        self._ck(self._lib.rdc_assemble_ripf(self._h, C.byref(p)))

    def assemble_hcc(self, p: HccParams):
        self._ck(self._lib.rdc_assemble_hcc(self._h, C.byref(p)))

    def assemble_proteas(self, p):
        """assemble_proteas_model (src/proteas.C:338-705); FIELD_AUX_NODAL = {HU, RTD, 0} per node."""
        self._ck(self._lib.rdc_assemble_proteas(self._h, C.byref(p)))

    def assemble_adpm(self, p):
        """assemble_adpm (src/adpm.C:324-652); the tract vectors go into FIELD_ELEM_TRACTS first."""
        self._ck(self._lib.rdc_assemble_adpm(self._h, C.byref(p)))

    def assemble_pihna_part(self, p: PihnaParams, part, stream=0):
        """one part (1 | 2, 0 = whole) of a two-part step on the given hipStream_t, in one C-ABI call"""
        self._ck(self._lib.rdc_assemble_pihna_part(self._h, C.byref(p), int(part), C.c_void_p(int(stream) or None)))

    def assemble_hcc_part(self, p: HccParams, part, stream=0):
        self._ck(self._lib.rdc_assemble_hcc_part(self._h, C.byref(p), int(part), C.c_void_p(int(stream) or None)))

    def solid_assemble_part(self, p: SolidParams, request_jacobian, part, stream=0):
        self._ck(self._lib.rdc_solid_assemble_part(self._h, C.byref(p), 1 if request_jacobian else 0, int(part), C.c_void_p(int(stream) or None)))

    def solid_assemble(self, p: SolidParams, request_jacobian=True):
        self._ck(self._lib.rdc_solid_assemble(self._h, C.byref(p), 1 if request_jacobian else 0))

    # -- results
    def csr_values_device_ptr(self):
        v, r = C.c_void_p(), C.c_void_p()
        self._ck(self._lib.rdc_csr_values_device_ptr(self._h, C.byref(v), C.byref(r)))
        return v.value, r.value

    def csr_download(self, want_val=True, want_rhs=True):
        n_rows, nnz = self.csr_dims()
        val = np.empty(nnz, dtype=np.float64) if want_val else None
        rhs = np.empty(n_rows, dtype=np.float64) if want_rhs else None
        self._ck(self._lib.rdc_csr_download(self._h, _dp(val) if want_val else None, _dp(rhs) if want_rhs else None))
        return val, rhs

    def csr_download_rows(self, node_begin, node_end, val_ptr, rhs_ptr, asynchronous=False):
        """rows of nodes [node_begin, node_end) into full-size host arrays given by ADDRESS (e.g. pinned torch tensors)"""
        self._ck(self._lib.rdc_csr_download_rows(self._h, int(node_begin), int(node_end), C.c_void_p(int(val_ptr) if val_ptr else None),
                                                 C.c_void_p(int(rhs_ptr) if rhs_ptr else None), 1 if asynchronous else 0))

    def part1_nodes(self):
        n = C.c_int64()
        self._ck(self._lib.rdc_part1_nodes(self._h, C.byref(n)))
        return n.value

    # -- instrumentation
    def timing_enable(self, on=True):
        self._ck(self._lib.rdc_timing_enable(self._h, 1 if on else 0))

    def timing_last_ms(self):
        ms = C.c_float()
        self._ck(self._lib.rdc_timing_last_ms(self._h, C.byref(ms)))
        return ms.value

    def timing_sum_ms(self):
        """(total device ms, number of assemble calls) since the last enable / sum; resets the pool."""
        ms, n = C.c_float(), C.c_int()
        self._ck(self._lib.rdc_timing_sum_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def timing_samples_ms(self, capacity=4096):
        """device ms of every assemble call since the last enable / sum / samples (oldest first); resets the pool."""
        buf = (C.c_float * capacity)()
        n = C.c_int()
        self._ck(self._lib.rdc_timing_samples_ms(self._h, buf, capacity, C.byref(n)))
        return [buf[i] for i in range(min(n.value, capacity))]
