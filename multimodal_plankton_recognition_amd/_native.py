"""ctypes binding of libmpr_hip.so (the C-ABI library of hand-written gfx950 kernels).

``include/mpr_hip.h`` is the single source of truth: the prototypes are parsed from it, so every
declared entry point is bound (and a missing export fails at import of this module).  There is NO
fallback: if the library is not built, or a call is made with a non-device tensor, this raises.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get('MPR_HIP_LIB') or os.path.join(_HERE, 'libmpr_hip.so')     # (override: A/B of two builds)
HEADER_PATH = os.path.join(_ROOT, 'include', 'mpr_hip.h')
DEBUG_HEADER_PATH = os.path.join(_ROOT, 'include', 'mpr_hip_debug.h')      # tuning knobs / timing hooks (not the boundary)


class NativeLibraryError(RuntimeError):
    pass


def _ctype(decl):
    decl = decl.strip()
    if '*' in decl:
        return ctypes.c_void_p
    base = re.sub(r'\b(const|restrict)\b', '', decl)
    base = re.sub(r'\b\w+$', '', base.strip()).strip() if len(base.split()) > 1 else base.strip()
    return {'int': ctypes.c_int, 'long long': ctypes.c_longlong, 'float': ctypes.c_float,
            'unsigned': ctypes.c_uint, 'unsigned int': ctypes.c_uint}[base]


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    protos = {}
    for m in re.finditer(r'(?:^|\n)\s*((?:const\s+)?\w[\w\s]*?\*?)\s*(mpr_\w+)\s*\(([^;{]*?)\)\s*;', text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if '*' in ret else (None if ret == 'void' else ctypes.c_longlong if ret == 'long long' else ctypes.c_int)
        if args in ('void', ''):
            argtypes = []
        elif '...' in args:
            argtypes = None   # variadic: not called from Python
        else:
            argtypes = [_ctype(a) for a in args.split(',')]
        protos[name] = (restype, argtypes)
    return protos


_lib = None
_protos = None


def lib():
    global _lib, _protos
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'(or `make -C multimodal_plankton_recognition_amd/csrc`). There is no CPU fallback.')
        handle = ctypes.CDLL(LIB_PATH)
        _protos = parse_header()
        _protos.update(parse_header(DEBUG_HEADER_PATH))
        for name, (restype, argtypes) in _protos.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise NativeLibraryError(f'{LIB_PATH} does not export {name} declared in mpr_hip.h') from e
            fn.restype = restype
            if argtypes is not None:
                fn.argtypes = argtypes
        _lib = handle
    return _lib


def exported_symbols():
    lib()
    return sorted(_protos)


def last_error():
    return lib().mpr_last_error().decode()


def is_dense(t):
    """Every element of the spanned storage used exactly once: a contiguous tensor or any permutation of one
    (e.g. a channels-last conv weight, whose [K][R][S][C] memory the conv kernels address through strides)."""
    if t.is_contiguous() or t.numel() <= 1:
        return True
    expect = 1
    for st, sz in sorted((st, sz) for st, sz in zip(t.stride(), t.shape) if sz > 1):
        if st != expect:
            return False
        expect *= sz
    return True


def ptr(t):
    """Device pointer of a dense CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise NativeLibraryError('HIP kernels need device tensors (got a CPU tensor); there is no CPU fallback')
    if not is_dense(t):
        raise NativeLibraryError('HIP kernels need contiguous tensors')
    return t.data_ptr()


def stream(device_index=None):
    """Raw hipStream_t of torch's current stream (the C-level getter: the Python Stream object costs ~8 us)."""
    if device_index is None:
        device_index = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(device_index)


def call(name, *args, stream_handle=None):
    """Invoke an int-returning entry point on torch's current stream (or the given raw hipStream_t); tensors become
    device pointers."""
    fn = getattr(lib(), name)
    conv = []
    dev = None
    for a in args:
        if a is None:
            conv.append(None)
        elif torch.is_tensor(a):
            conv.append(ptr(a))
            if dev is None:
                dev = a.device.index
        else:
            conv.append(a)
    rc = fn(*conv, stream(dev) if stream_handle is None else stream_handle)
    if rc != 0:
        raise NativeLibraryError(f'{name} failed (rc={rc}): {last_error()}')


def query(name, *args):
    """Invoke a pure host-side size query (no stream argument) and return its int result."""
    return getattr(lib(), name)(*args)
