# -*- coding: utf-8 -*-
"""ctypes binding of the C ABI declared in include/oriana_hip.h.

The HIP library is the product path: if it cannot be loaded this module raises -- there is no
CPU fallback (the CPU restatement under oracle/ is test infrastructure only).
"""
import ctypes
import os
from ctypes import c_int, c_int64, c_void_p, c_double, c_char_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ORIANA_HIP_LIB') or os.path.join(HERE, 'csrc', 'liboriana_hip.so')   # env: analysis builds


class OrianaHipError(RuntimeError):
    pass


class OrianaCounts(ctypes.Structure):
    """struct oriana_counts (include/oriana_hip.h)."""
    _fields_ = [('n', c_int64), ('m', c_int64), ('nrb', c_int64), ('ncb', c_int64), ('nnz', c_int64),
                ('rslots', c_int64), ('cslots', c_int64),
                ('roff', c_void_p), ('coff', c_void_p), ('rslice', c_void_p), ('cslice', c_void_p),
                ('rowrec', c_void_p), ('ridx', c_void_p), ('col_perm', c_void_p), ('row_perm', c_void_p)]


class OrianaDense(ctypes.Structure):
    """struct oriana_dense (include/oriana_hip.h)."""
    _fields_ = [('n', c_int64), ('gd', c_int64), ('nct', c_int64), ('x', c_void_p)]


class OrianaRowSplit(ctypes.Structure):
    """struct oriana_row_split (include/oriana_hip.h): which work-groups of the row pass share a row block."""
    _fields_ = [('nfull', ctypes.c_int32), ('parts', ctypes.c_int32), ('edge', ctypes.c_int32 * 9)]


class OrianaClearList(ctypes.Structure):
    """struct oriana_clear_list (include/oriana_hip.h): up to 8 buffers zero-filled by the factor preparation's launch."""
    _fields_ = [('ptr', c_void_p * 8), ('bytes', c_int64 * 8)]


_P = c_void_p
_I = c_int64
_SIGS = {
    'oriana_kpad': (c_int64, [_I]),
    'oriana_col_block_tiles': (c_int64, [_I]),
    'oriana_version': (c_char_p, []),
    'oriana_pack_count': (c_int, [_P, c_int, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    'oriana_pack_fill': (c_int, [_P, c_int, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    'oriana_factor_prep': (c_int, [_P, _P, _P, _P, _P, _I, _I, _P]),
    'oriana_factor_prep_pair': (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    'oriana_factor_prep_pair_clear': (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, ctypes.POINTER(OrianaClearList), _P]),
    'oriana_factor_prep_pair_fused': (c_int, [_P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _P, ctypes.POINTER(OrianaClearList), _P]),
    'oriana_prep_scratch_bytes': (_I, []),
    'oriana_row_pass': (c_int, [ctypes.POINTER(OrianaCounts), _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    'oriana_row_pass_gene_splits': (_I, [ctypes.POINTER(OrianaCounts), _I]),
    'oriana_row_pass_split': (c_int, [ctypes.POINTER(OrianaCounts), _P, _P, _P, _P, _P, _I, _I, _P]),
    'oriana_row_spmm': (c_int, [ctypes.POINTER(OrianaCounts), _P, _P, _P, _P, _I, _P]),
    'oriana_row_pass_masked': (c_int, [ctypes.POINTER(OrianaCounts), _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    'oriana_row_pass_general': (c_int, [ctypes.POINTER(OrianaCounts), _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, ctypes.POINTER(OrianaRowSplit), _P, _P]),
    'oriana_prep_den_threshold_offset': (_I, []),
    'oriana_row_pass_plan': (c_int, [ctypes.POINTER(OrianaCounts), _I, _P, ctypes.POINTER(OrianaRowSplit)]),
    'oriana_row_pass_plan_cus': (c_int, [ctypes.POINTER(OrianaCounts), _I, _P, _I, ctypes.POINTER(OrianaRowSplit)]),
    'oriana_device_cus': (_I, []),
    'oriana_plan_gene_order': (c_int, [_P, _P, _I, _I, c_double, c_double, _P, _P]),
    'oriana_plan_col_work_capacity': (_I, [_I, _I, _I]),
    'oriana_plan_col_work': (c_int, [_P, _I, _I, _I, _I, _I, c_int, c_int, _P, _I, _P]),
    'oriana_plan_dense_splits': (c_int, [_I, _I, _I, _P, _P]),
    'oriana_counts_create_dense_f32': (c_int, [_P, _P, _I, _I, _I, _I, c_double, _P]),
    'oriana_counts_create_csr': (c_int, [_P, _P, _P, _P, _I, _I, _I, c_double, _P]),
    'oriana_counts_destroy': (c_int, [_P]),
    'oriana_counts_declare_unit_dropout': (c_int, [_P, c_int]),
    'oriana_counts_info': (c_int, [_P, _P, _I]),
    'oriana_zq_gap_resident': (c_int, [_P, _P, _P, _P, _P, _P]),
    'oriana_zq_zigap_resident': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, _P]),
    'oriana_zq_sparse_gap_resident': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'oriana_zq_sparse_zigap_resident': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'oriana_col_pass': (c_int, [ctypes.POINTER(OrianaCounts), _P, _P, _P, _I, _P, _I, _P]),
    'oriana_col_pass_dual': (c_int, [ctypes.POINTER(OrianaCounts), _P, _P, _P, _P, _P, _I, _P, _I, _P]),
    'oriana_col_pass_f64acc': (c_int, [ctypes.POINTER(OrianaCounts), _P, _P, _P, _I, _P]),
    'oriana_col_pass_det_scratch_bytes': (c_int64, [_I, _I]),
    'oriana_col_pass_det': (c_int, [ctypes.POINTER(OrianaCounts), _P, _P, _P, _I, _P, _I, _P, _P]),
    'oriana_dense_supported': (c_int, [_I]),
    'oriana_dense_image_pieces': (c_int64, [_I, c_int]),
    'oriana_dense_pack': (c_int, [_P, c_int, _I, _I, _I, _I, _P, _P]),
    'oriana_dense_images': (c_int, [_P, _P, _I, _I, c_int, _P]),
    'oriana_dense_row_pass': (c_int, [ctypes.POINTER(OrianaDense), _P, _P, _P, _P, _P, _I, _I, _P]),
    'oriana_dense_row_pass_tail': (c_int, [ctypes.POINTER(OrianaDense), _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    'oriana_dense_col_pass': (c_int, [ctypes.POINTER(OrianaDense), _P, _P, _P, _I, _I, _P]),
    'oriana_dense_fixup': (c_int, [ctypes.POINTER(OrianaDense), _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    'oriana_dense_metric': (c_int, [ctypes.POINTER(OrianaDense), _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    'oriana_dense_fixup_weighted': (c_int, [ctypes.POINTER(OrianaDense), _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    'oriana_dense_fixup_variant': (c_int, [ctypes.POINTER(OrianaDense), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, c_int, _P]),
    'oriana_dense_images2': (c_int, [_P, _P, _P, _I, _I, c_int, _P]),
    'oriana_dense_fix_nz': (c_int, [ctypes.POINTER(OrianaDense), _P, _I, _P, _P, c_double, _P]),
    'oriana_finalize': (c_int, [_P, _P, _P, _P, _P, _I, _I, c_int, _P]),
    'oriana_finalize_slabs': (c_int, [_P, _P, _P, _I, _P, _I, _I, _P]),
    'oriana_finalize_slabs_from': (c_int, [_P, _P, _P, _I, _I, _P, _I, _I, _P]),
    'oriana_fixup': (c_int, [ctypes.POINTER(OrianaCounts), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                             _I, c_int, _P]),
    'oriana_zq_workspace_bytes': (c_int64, [_I, _I, _I, _I]),
    'oriana_zq_gap_f32': (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _I, _P]),
    'oriana_zq_zigap_f32': (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, c_int, _P, _I, _P]),
    'oriana_zq_sparse_gap_f32': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _I, _P]),
    'oriana_zq_sparse_zigap_f32': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _I, _P]),
    'oriana_gamma_update': (c_int, [_P] * 13 + [_I, _I, _P]),
    'oriana_mstep_gamma': (c_int, [_P, _P, _P, _P, c_double, _I, _P]),
    'oriana_gamma_update_finalize': (c_int, [_P] * 11 + [_I, _P, _P, _I, _I, _P]),
    'oriana_gamma_update_finalize_from': (c_int, [_P] * 11 + [_I, _I, _P, _P, _I, _I, _P]),
    'oriana_gamma_update_prep_blocks': (_I, [_I, _I]),
    'oriana_gamma_update_prep': (c_int, [_P] * 13 + [_I, _I, _P, _P, _P, _P]),
    'oriana_gamma_update_finalize_prep': (c_int, [_P] * 11 + [_I, _I, _P, _P, _I, _I, _P, _P, _P, _P]),
    'oriana_gamma_update_finalize_lazy': (c_int, [_P] * 10 + [_I, _I, _P, _P, _I, _I, _P, _P, _P, _P]),
    'oriana_mstep_gamma_pair': (c_int, [_P, _P, _P, _P, c_double, _P, _P, _P, _P, c_double, _P, _I, _P]),
    'oriana_colsum_f64': (c_int, [_P, _P, _P, _I, _I, _P]),
    'oriana_dropout_update': (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _P]),
    'oriana_dropout_update_fused': (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'oriana_dense_times_factor': (c_int, [_P, _P, _P, _I, _I, _I, c_int, _P]),
    'oriana_dropout_sweep_fused': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _I, _I, _I, _P]),
    'oriana_dropout_sweep_scratch_floats': (_I, [_I, _I]),
    'oriana_dropout_sweep_fused_tiles': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _I, _I, _I, _P]),
    'oriana_nzmask_tiles_words': (_I, [_I, _I]),
    'oriana_nzmask_tiles': (c_int, [_P, _P, _I, _I, _P]),
    'oriana_dense_t_times_factor_f32': (c_int, [_P, _P, _P, _P, c_int, _I, _I, _I, _P]),
    'oriana_dense_t_scratch_floats': (_I, [_I, _I]),
    'oriana_factor_cast_f32': (c_int, [_P, _P, _P, _P, _I, _I, _P]),
    'oriana_metric_nnz': (c_int, [ctypes.POINTER(OrianaCounts), _P, _P, _P, _I, _P, _P]),
    'oriana_count_stats': (c_int, [ctypes.POINTER(OrianaCounts), _P, _P, _P, _P]),
    'oriana_dropout_metric': (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    'oriana_nzmask_f32': (c_int, [_P, _P, _I, _I, _P]),
    'oriana_dropout_fix_nz': (c_int, [ctypes.POINTER(OrianaCounts), _P, _P, c_double, _P]),
    'oriana_dropout_fix_nz_ld': (c_int, [ctypes.POINTER(OrianaCounts), _P, _P, c_double, _I, _P]),
    'oriana_mul_f64_f32': (c_int, [_P, _P, _P, _I, _P]),
    'oriana_colsum_wide_f64': (c_int, [_P, _P, _I, _I, _P]),
    'oriana_colsum_wide_f32': (c_int, [_P, _P, _I, _I, _P]),
    'oriana_take_cols_f32': (c_int, [_P, _P, _I, _I, _I, _P]),
    'oriana_sparsity_update': (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    'oriana_threshold_f32': (c_int, [_P, _P, c_double, _I, _P]),
    'oriana_rowmean_f64': (c_int, [_P, _P, _I, _I, _P]),
    'oriana_scale_factor': (c_int, [_P, _P, _P, _P, _I, _I, c_int, _P]),
    'oriana_finalize_zlog': (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    'oriana_log_center': (c_int, [_P, _P, _P, _P, _P, _I, _I, _P]),
    'oriana_scale_factor_centered': (c_int, [_P, _P, _P, _P, _P, _I, _I, _P]),
    'oriana_prep_center_offset': (_I, []),
    'oriana_digamma_f64': (c_int, [_P, _P, _I, _P]),
    'oriana_trigamma_f64': (c_int, [_P, _P, _I, _P]),
    'oriana_inverse_digamma_f64': (c_int, [_P, _P, _I, _P]),
    'oriana_sigmoid_f64': (c_int, [_P, _P, _I, _P]),
    'oriana_logit_f64': (c_int, [_P, _P, _I, _P]),
}

_lib = None


def load():
    """Load liboriana_hip.so (building it is __graft_entry__.build()'s job).  Raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OrianaHipError(
                'HIP library %s is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                '(hipcc --offload-arch=gfx950).  There is no CPU fallback.' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            if not hasattr(lib, name):
                continue              # symbol table is checked separately (tests/test_abi.py)
            f = getattr(lib, name)
            f.restype = res
            f.argtypes = args
        _lib = lib
    return _lib


def declared_symbols():
    return sorted(_SIGS)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


_raw_stream = None


def stream_ptr():
    """The HIP stream torch currently launches on (of the current device), as an integer handle.  Called once per
    launch: torch.cuda.current_stream() builds a Stream object and resolves the device index in Python (8 us -- more
    than a small kernel runs); the raw getter behind it is a single C call."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', False)
    if _raw_stream:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Call an int-returning entry point; non-zero return -> OrianaHipError."""
    f = getattr(load(), name)
    rc = f(*args)
    if rc != 0:
        raise OrianaHipError('%s failed with code %d' % (name, rc))
    return rc
