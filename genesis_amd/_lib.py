"""ctypes binding of libgenesis_hip.so.  Prototypes are parsed from include/genesis_hip.h so the
header is the single source of truth for the C ABI.  There is no fallback: if the library is
missing, or a call returns an error code, this raises."""
import ctypes
import os
import os.path as osp
import re

# torch bundles its own libamdhip64.so.7; import it FIRST so that libgenesis_hip.so (same SONAME dependency)
# binds to the HIP runtime torch's streams and allocations live in, not to a second copy from /opt/rocm.
import torch  # noqa: F401

_HERE = osp.dirname(osp.abspath(__file__))
# GENESIS_HIP_LIB: an alternative build of the same library (kernel experiments: tools/abl_build.sh)
LIB_PATH = os.environ.get('GENESIS_HIP_LIB') or osp.join(_HERE, 'libgenesis_hip.so')
HEADER_PATH = osp.join(osp.dirname(_HERE), 'include', 'genesis_hip.h')

_CTYPES = {
    'int': ctypes.c_int, 'float': ctypes.c_float, 'double': ctypes.c_double, 'size_t': ctypes.c_size_t,
    'gx_stream_t': ctypes.c_void_p, 'void': None, 'long long': ctypes.c_longlong, 'unsigned long long': ctypes.c_ulonglong,
}


class GenesisHipError(RuntimeError):
    pass


def parse_header(path=HEADER_PATH):
    """-> {name: (restype_str, [(type_str, argname), ...])} for every gx_* declaration."""
    text = open(path).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    text = re.sub(r'//[^\n]*', '', text)
    text = '\n'.join(l for l in text.splitlines() if not l.strip().startswith('#'))
    decls = {}
    for m in re.finditer(r'([\w\s\*]+?)\b(gx_\w+)\s*\(([^)]*)\)\s*;', text):
        ret = ' '.join(m.group(1).replace('extern "C"', '').split())
        args = []
        body = m.group(3).strip()
        if body and body != 'void':
            for a in body.split(','):
                a = ' '.join(a.split())
                mm = re.match(r'(.*?)(\w+)$', a)
                args.append((mm.group(1).strip(), mm.group(2)))
        decls[m.group(2)] = (ret, args)
    return decls


def _ctype(tstr):
    t = tstr.replace('const', '').strip()
    if t.endswith('*'):
        return ctypes.c_char_p if t.replace(' ', '') == 'char*' else ctypes.c_void_p
    return _CTYPES[t]


_lib = None
_decls = None


def load():
    global _lib, _decls
    if _lib is not None:
        return _lib
    if not osp.exists(LIB_PATH):
        raise GenesisHipError(
            'libgenesis_hip.so not built (%s). Run `python -m genesis_amd.build`; the HIP hot path has '
            'no CPU or eager fallback.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    _decls = parse_header()
    for name, (ret, args) in _decls.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = _ctype(ret) if ret != 'void' else None
        fn.argtypes = [_ctype(t) for t, _ in args]
    _lib = lib
    return lib


def declarations():
    load()
    return _decls


def last_error():
    return load().gx_last_error().decode()


def check(rc, name):
    if rc != 0:
        raise GenesisHipError('%s failed (%d): %s' % (name, rc, last_error()))


def call(name, *args):
    """Calls an int-returning gx_* entry point and raises on a non-zero code."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise GenesisHipError('%s failed (%d): %s' % (name, rc, last_error()))


def query(name, *args):
    """Calls a size_t-returning *_ws_bytes query."""
    return int(getattr(load(), name)(*args))


def current_ctx():
    """Id of the calling thread's current library context (gx_ctx_*: include/genesis_hip.h)."""
    return int(load().gx_ctx_current())


def make_current(ctx_id):
    call('gx_ctx_make_current', int(ctx_id))
