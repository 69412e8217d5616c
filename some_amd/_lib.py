"""ctypes binding of libsome_amd.so - the only door between the Python host side and the HIP kernels.

The library is built in-tree by ``python -m some_amd.build`` (or ``__graft_entry__.build()``).  There is NO
fallback: if the shared object is missing or a call fails, the Python layer raises.
"""
import ctypes as C
import pathlib

import os

_HERE = pathlib.Path(__file__).resolve().parent
# SOME_AMD_LIBRARY: an alternative build of the same library (kernel A/B experiments, tools/build_variant.py)
LIB_PATH = pathlib.Path(os.environ['SOME_AMD_LIBRARY']).resolve() if os.environ.get('SOME_AMD_LIBRARY') else _HERE / 'libsome_amd.so'

SOME_OK = 0
SOME_EINVAL, SOME_EKEY, SOME_ESHAPE, SOME_EHIP, SOME_ESTATE, SOME_ENOMEM = -1, -2, -3, -4, -5, -6
HEAD_LOGITS, HEAD_SIGMOID, HEAD_SOFTMAX = 0, 1, 2
EPI_NONE, EPI_BIAS, EPI_BIAS_SILU, EPI_BIAS_RES, EPI_GLU, EPI_GLU_RES = range(6)
PRECISION_F32, PRECISION_F16X3, PRECISION_F16X3_FAST = 0, 1, 2
PAD_ZERO, PAD_REFLECT = 0, 1
SAMPLE_F32, SAMPLE_PCM16 = 0, 1
(ELT_SILU_FWD, ELT_SILU_BWD, ELT_SIGMOID_FWD, ELT_SIGMOID_BWD, ELT_AXPY, ELT_DROPOUT, ELT_SILU_DROP_FWD, ELT_SILU_DROP_BWD,
 ELT_AXPY_DROP) = range(9)
GEMM_SPLIT_IN, GEMM_SPLIT_OUT, GEMM_HI_ONLY, GEMM_HI_BF16 = 1, 2, 4, 8
OPERAND_F16X2, OPERAND_BF16 = 0, 1


class SomeConfig(C.Structure):
    _fields_ = [
        ('lay', C.c_int32), ('dim', C.c_int32), ('heads', C.c_int32), ('head_dim', C.c_int32),
        ('kernel_size', C.c_int32), ('indim', C.c_int32), ('outdim', C.c_int32),
        ('sample_rate', C.c_int32), ('hop_size', C.c_int32), ('win_size', C.c_int32),
        ('fmin', C.c_float), ('fmax', C.c_float),
        ('midi_min', C.c_double), ('midi_max', C.c_double),
        ('midi_deviation', C.c_double), ('rest_threshold', C.c_double),
        ('precision', C.c_int32), ('reserved', C.c_int32),
    ]


class SomeTensorDesc(C.Structure):
    _fields_ = [('name', C.c_char_p), ('data', C.c_void_p), ('dtype', C.c_int32), ('ndim', C.c_int32),
                ('shape', C.c_int64 * 4)]


class SomeKernelStat(C.Structure):
    _fields_ = [('name', C.c_char * 48), ('launches', C.c_int64), ('total_ms', C.c_double),
                ('flops', C.c_double), ('bytes', C.c_double)]


# every symbol include/some_amd.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    'some_create': (C.c_int, [C.POINTER(SomeConfig), C.POINTER(_P)]),
    'some_destroy': (None, [_P]),
    'some_last_error': (C.c_char_p, [_P]),
    'some_version': (C.c_char_p, []),
    'some_arena_bytes': (C.c_size_t, [_P]),
    'some_pack_weights': (C.c_int, [_P, C.POINTER(SomeTensorDesc), C.c_int32, _P]),
    'some_attach_arena': (C.c_int, [_P, _P, C.c_size_t]),
    'some_mel_filterbank': (C.c_int, [_P, _P]),
    'some_logmel': (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    'some_logmel_shifted': (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int32, _P, _P]),
    'some_workspace_bytes': (C.c_size_t, [_P, C.c_int64, C.c_int32]),
    'some_forward': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int64, C.c_int32, _P, C.c_int32, _P, _P, _P, C.c_size_t, _P]),
    'some_decode_scratch_bytes': (C.c_size_t, [_P, C.c_int64]),
    'some_decode': (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int64, C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    'some_decode_notes': (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    'some_slicer_rms': (C.c_int, [_P, _P, C.c_int32, _P, _P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, _P, _P]),
    'some_pcm_gather': (C.c_int, [_P, _P, C.c_int32, _P, _P, C.c_int32, C.c_int64, _P, _P]),
    'some_train_scratch_bytes': (C.c_size_t, [_P, C.c_int64, C.c_int32]),
    'some_train_gemm_splitk_bytes': (C.c_size_t, [_P, C.c_int32, C.c_int32, C.c_int32]),
    'some_train_gemm_splitk': (C.c_int, [_P, _P, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_size_t, _P]),
    'some_train_gemm16_bytes': (C.c_size_t, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'some_train_gemm16': (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_int32, _P, C.c_size_t, _P]),
    'some_train_gemm16_wgrad': (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_size_t, _P]),
    'some_train_transpose': (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, _P]),
    'some_train_cast16': (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, _P]),
    'some_train_silu16': (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, _P]),
    'some_train_transpose16': (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    'some_train_transpose16_table': (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    'some_train_gemm16s': (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, C.c_int32, _P, _P, C.c_int32, _P, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_float, C.c_uint64, C.c_float, _P]),
    'some_train_dropcast16': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_uint64, C.c_int32, _P]),
    'some_train_layernorm_fwd16': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P]),
    'some_train_ffn_block_save_bytes': (C.c_size_t, [_P, C.c_int32, C.c_int32, C.c_int32]),
    'some_train_ffn_block_scratch_bytes': (C.c_size_t, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'some_train_ffn_block_fwd': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                           C.c_uint64, C.c_float, C.c_uint64, _P, C.c_size_t, _P, _P]),
    'some_train_ffn_block_bwd': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                           C.c_uint64, C.c_float, C.c_uint64, _P, _P, _P, _P, _P, _P, C.c_int32, _P, _P, C.c_size_t, _P, C.c_size_t,
                                           _P, C.c_size_t, _P]),
    'some_train_layernorm_bwd_add': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P, C.c_size_t, _P]),
    'some_train_gemm16_wgrad16': (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P,
                                            C.c_size_t, _P]),
    'some_train_set_wgrad_stream': (C.c_int, [_P, _P, _P, C.c_int32]),
    'some_train_wgrad_flush': (C.c_int, [_P, _P]),
    'some_train_weighted_colsum': (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_size_t, _P]),
    'some_train_colsum': (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, C.c_size_t, _P]),
    'some_train_layernorm_fwd': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int32, _P]),
    'some_train_layernorm_bwd': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P, C.c_size_t, _P]),
    'some_train_batchnorm_fwd': (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_float, C.c_float, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    'some_train_batchnorm_bwd': (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_size_t, _P]),
    'some_train_eltwise': (C.c_int, [_P, C.c_int32, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_uint64, _P]),
    'some_train_glu': (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, _P]),
    'some_train_mask_rows': (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, _P]),
    'some_train_dwconv': (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, _P]),
    'some_train_dwconv_bwd_taps': (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P, C.c_int32, _P, C.c_size_t, _P]),
    'some_train_dwconv_bwd_params': (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P, _P, _P, C.c_size_t, _P]),
    'some_train_bce_with_logits': (C.c_int, [_P, _P, _P, C.c_int64, _P, _P, _P, C.c_size_t, _P]),
    'some_train_cross_entropy': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int64, _P, _P, _P, C.c_size_t, _P]),
    'some_train_binary_emd': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P, _P, _P, C.c_size_t, _P]),
    'some_train_sumsq': (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_size_t, _P]),
    'some_train_adamw': (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_float, _P]),
    'some_train_adamw_clip': (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, _P, C.c_double,
                                       C.c_double, _P]),
    'some_train_attention_fwd': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    'some_train_attention_bwd': (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    'some_train_attention_fwd_f16x3': (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    'some_train_attention_bwd_f16x3': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    'some_train_split_transpose': (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_int32, _P]),
    'some_train_attention_bwd16_work_bytes': (C.c_size_t, [_P, C.c_int32, C.c_int32]),
    'some_train_attention_bwd_f16x3_auto16': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_size_t,
                                                        _P]),
    'some_train_attention_bwd_f16x3_out16': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P,
                                                       _P]),
    'some_op_gemm': (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, _P, _P, C.c_int32, _P, C.c_int32,
                               C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, _P, C.c_int32, _P]),
    'some_op_split_rows': (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, _P]),
    'some_op_split_rows_fmt': (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_int32, _P]),
    'some_op_layernorm': (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, _P]),
    'some_op_attention': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P, C.c_int32, _P]),
    'some_op_qkv_attention_f16x3_bytes': (C.c_size_t, [C.c_int32, C.c_int32]),
    'some_op_qkv_attention_f16x3': (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_size_t, _P]),
    'some_op_dwconv_silu': (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P, C.c_int32, _P]),
    'some_profile_enable': (C.c_int, [_P, C.c_int32]),
    'some_profile_collect': (C.c_int, [_P, C.POINTER(SomeKernelStat), C.c_int32, C.POINTER(C.c_int32)]),
    'some_box_calibrate': (C.c_int, [C.c_double, _P, _P, _P, _P]),
}

_lib = None


class SomeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f'libsome_amd error {code}: {msg}')
        self.code = code


def load():
    """Load (once) and return the ctypes library with typed signatures.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: libsome_amd.so must bind to the SAME HIP runtime (libamdhip64.so.7) that PyTorch-ROCm
    # bundles and initialises, because every device pointer it receives comes from torch's allocator.  A
    # process that dlopens the library before torch would pull in /opt/rocm's copy as a second runtime.
    import torch  # noqa: F401
    if not LIB_PATH.exists():
        raise RuntimeError(
            f'{LIB_PATH} is missing: the HIP extension has not been built. Run `python -m some_amd.build` '
            f'(needs hipcc, ROCm >= 7.0). There is no CPU / PyTorch fallback for the hot path.')
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)     # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


_fast = None


def fast():
    """The library for the call-heavy paths (the training operators): a namespace with every entry point of ``SYMBOLS`` - the generated
    METH_FASTCALL wrapper (some_amd/_fastcall.so, some_amd/fastcall_gen.py: 0.3 us per call instead of ctypes' ~2 us for 20 arguments)
    where the signature is plain pointers / integers / floats, the ctypes function otherwise.  Same library, same entry points, same
    argument order; the wrappers take pointers as None / int / ctypes objects.  SOME_AMD_FASTCALL=0 (A/B runs) or a tree without the
    built module (it is a binding, not a compute path): the ctypes library itself."""
    global _fast
    if _fast is not None:
        return _fast
    lib = load()
    if os.environ.get('SOME_AMD_FASTCALL', '1') == '0' or not (_HERE / '_fastcall.so').exists():
        _fast = lib
        return lib
    import types
    try:
        from . import _fastcall
    except ImportError as e:                      # built for another interpreter: the binding is optional, the library is not
        import warnings
        warnings.warn(f'some_amd/_fastcall.so does not import ({e}); using the ctypes binding (rebuild: python -m some_amd.build)')
        _fast = lib
        return lib
    from .fastcall_gen import table_digest
    built_from = _fastcall.digest() if hasattr(_fastcall, 'digest') else None
    if built_from != table_digest(SYMBOLS):       # generated from another SYMBOLS table: an argument of another width would be marshalled wrongly
        import warnings
        warnings.warn('some_amd/_fastcall.so was generated from a different _lib.SYMBOLS table; using the ctypes binding '
                      '(rebuild: python -m some_amd.build)')
        _fast = lib
        return lib
    ns = types.SimpleNamespace()
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if _fastcall.bind(name, C.cast(fn, C.c_void_p).value):
            fn = getattr(_fastcall, name)
        setattr(ns, name, fn)
    ns.fastcall = True
    _fast = ns
    return ns


def check(handle, rc):
    if rc != SOME_OK:
        msg = load().some_last_error(handle)
        raise SomeError(rc, msg.decode('utf8', 'replace') if msg else '')
