"""ctypes binding of the C ABI in include/mimir_gpu_shuffle.h.

The library is the product: there is no Python or CPU fallback.  Loading fails
loudly when libmimir_gpu_shuffle.so has not been built
(`python -m mimir_amd.build`), and every non-zero status raises `MgsError`.
"""
import ctypes as C
import os

from . import build as _build

MGS_MAX_RANKS = 256

KEY_FIXED, KEY_CSTR = 0, 1
VAL_VOID, VAL_FIXED, VAL_CSTR = 0, 1, 2
ALGO_AUTO, ALGO_GENERIC, ALGO_ONESWEEP16, ALGO_ONESWEEPV = 0, 1, 2, 3
COMBINE_SUM_U64, COMBINE_MIN_U64, COMBINE_MAX_U64 = 0, 1, 2
FLAG_OSV_ROUNDS_KERNEL, FLAG_OSV_IMAGE_KERNEL, FLAG_OSV_STAGED_KERNEL = 1, 2, 4

OK = 0
ERR_ARG, ERR_HIP, ERR_RECORD_SIZE, ERR_LOCAL_OVERFLOW = -1, -2, -3, -4
ERR_MALFORMED, ERR_TIMEOUT, ERR_STATE = -5, -6, -7


class MgsError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"mgs status {status}: {message}")
        self.status = status


class Layout(C.Structure):
    _fields_ = [("key_kind", C.c_int32), ("key_bytes", C.c_int32),
                ("key_count", C.c_int32), ("val_kind", C.c_int32),
                ("val_bytes", C.c_int32), ("val_count", C.c_int32)]

    @classmethod
    def make(cls, key_kind, key_bytes, val_kind, val_bytes, key_count=1,
             val_count=1):
        return cls(key_kind, key_bytes, key_count, val_kind, val_bytes,
                   val_count)

    def fixed_len(self):
        if self.key_kind == KEY_FIXED and self.val_kind != VAL_CSTR:
            return self.key_bytes + (self.val_bytes
                                     if self.val_kind == VAL_FIXED else 0)
        return None


class Desc(C.Structure):
    _fields_ = [("nranks", C.c_int32), ("rank", C.c_int32),
                ("layout", Layout), ("buf_size", C.c_int64),
                ("device", C.c_int32), ("flags", C.c_int32),
                ("max_records", C.c_uint64)]


class PageCut(C.Structure):
    _fields_ = [("first_record", C.c_uint64), ("first_byte", C.c_uint64)]


class Round(C.Structure):
    _fields_ = [("next", C.c_uint64), ("packed_records", C.c_uint64),
                ("local_records", C.c_uint64), ("local_bytes", C.c_int64),
                ("send_offset", C.c_int64 * MGS_MAX_RANKS)]


class Rounds(C.Structure):
    _fields_ = [("n_rounds", C.c_uint32), ("reserved", C.c_uint32), ("next", C.c_uint64),
                ("packed_records", C.c_uint64), ("local_records", C.c_uint64),
                ("local_bytes", C.c_int64)]


# every symbol include/mimir_gpu_shuffle.h declares: (restype, argtypes)
_VP = C.c_void_p
SYMBOLS = {
    "mgs_geometry": (C.c_int, [C.c_int64, C.c_int, C.POINTER(C.c_int64),
                               C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mgs_hashlittle_host": (C.c_uint32, [_VP, C.c_size_t, C.c_uint32]),
    "mgs_dest_of_hash_host": (C.c_uint32, [C.c_uint32, C.c_uint32]),
    "mgs_create": (C.c_int, [C.POINTER(Desc), C.POINTER(_VP)]),
    "mgs_destroy": (None, [_VP]),
    "mgs_last_error": (C.c_char_p, [_VP]),
    "mgs_last_algo": (C.c_char_p, [_VP]),
    "mgs_last_kernel": (C.c_char_p, [_VP]),
    "mgs_pack_device": (C.c_int, [_VP, _VP, _VP, C.c_uint64, C.c_uint64, _VP,
                                  _VP, _VP, C.c_int64, _VP, C.c_int, _VP,
                                  C.POINTER(Round)]),
    "mgs_pack_device_async": (C.c_int, [_VP, _VP, _VP, C.c_uint64, C.c_uint64,
                                        _VP, _VP, _VP, C.c_int64, _VP, C.c_int,
                                        _VP]),
    "mgs_pack_result": (C.c_int, [_VP, C.POINTER(Round)]),
    "mgs_pack_rounds_device": (C.c_int, [_VP, _VP, _VP, C.c_uint64, C.c_uint64, _VP, _VP, C.c_int64,
                                         C.c_uint32, _VP, C.c_int64, _VP, C.c_int, _VP,
                                         C.POINTER(Rounds), _VP, _VP, _VP, _VP]),
    "mgs_pack_rounds_device_async": (C.c_int, [_VP, _VP, _VP, C.c_uint64, C.c_uint64, _VP, _VP, C.c_int64,
                                               C.c_uint32, _VP, C.c_int64, _VP, C.c_int, _VP]),
    "mgs_pack_rounds_result": (C.c_int, [_VP, C.POINTER(Rounds), _VP, _VP, _VP, _VP]),
    "mgs_pack_streams_device": (C.c_int, [_VP, _VP, _VP, C.c_uint64, C.c_uint64, _VP, _VP, _VP, C.c_int64,
                                          C.c_uint32, _VP, C.c_int64, _VP, _VP,
                                          C.POINTER(Rounds), _VP, _VP, _VP, _VP, _VP]),
    "mgs_pack_streams_device_async": (C.c_int, [_VP, _VP, _VP, C.c_uint64, C.c_uint64, _VP, _VP, _VP, C.c_int64,
                                                C.c_uint32, _VP, C.c_int64, _VP, _VP]),
    "mgs_pack_streams_result": (C.c_int, [_VP, C.POINTER(Rounds), _VP, _VP, _VP, _VP, _VP]),
    "mgs_pack_device_noalloc": (C.c_int, [_VP, _VP, _VP, C.c_uint64, C.c_uint64, _VP,
                                          _VP, _VP, C.c_int64, _VP, _VP,
                                          C.POINTER(Round)]),
    "mgs_last_kernel_ms": (C.c_int, [_VP, C.POINTER(C.c_float)]),
    "mgs_set_tuning": (C.c_int, [_VP, C.c_int]),
    "mgs_get_tuning": (C.c_int, [_VP]),
    "mgs_tuning_name": (C.c_char_p, [C.c_int]),
    "mgs_debug_stats": (C.c_int, [_VP, C.POINTER(C.c_uint64)]),
    "mgs_finder_stats": (C.c_int, [_VP, C.POINTER(C.c_uint64)]),
    "mgs_copy_calibration": (C.c_int, [_VP, _VP, _VP, C.c_uint64, _VP,
                                       C.POINTER(C.c_float)]),
    "mgs_get_desc": (C.c_int, [_VP, C.POINTER(Desc)]),
    "mgs_stager_create": (C.c_int, [_VP, C.c_int64, C.POINTER(_VP)]),
    "mgs_stager_destroy": (None, [_VP]),
    "mgs_stager_last_error": (C.c_char_p, [_VP]),
    "mgs_stager_put": (C.c_int, [_VP, _VP, C.c_uint32, _VP, C.c_uint32]),
    "mgs_stager_pending": (C.c_uint64, [_VP]),
    "mgs_stager_pack": (C.c_int, [_VP, C.POINTER(Round), C.POINTER(_VP),
                                  C.POINTER(C.c_int)]),
    "mgs_stager_send_image_host": (C.c_int, [_VP, C.POINTER(_VP),
                                             C.POINTER(C.POINTER(C.c_int64))]),
    "mgs_stager_send_image_device": (C.c_int, [_VP, C.POINTER(_VP),
                                               C.POINTER(C.POINTER(C.c_int64))]),
    "mgs_stager_round_done": (C.c_int, [_VP]),
    "mgs_stager_pack_begin": (C.c_int, [_VP]),
    "mgs_stager_pack_end": (C.c_int, [_VP, _VP, _VP, _VP]),
    "mgs_stager_send_image_to": (C.c_int, [_VP, _VP, _VP]),
    "mgs_host_alloc": (_VP, [C.c_size_t]),
    "mgs_host_free": (None, [_VP]),
    "mgs_device_alloc": (_VP, [C.c_int, C.c_size_t]),
    "mgs_device_free": (None, [C.c_int, _VP]),
    "mgs_copy_to_host": (C.c_int, [C.c_int, _VP, _VP, C.c_size_t]),
    "mgs_stager_window": (C.c_int, [_VP, _VP]),
    "mgs_stager_commit": (C.c_int, [_VP, C.c_int64, C.c_uint64]),
    "mgs_stager_recv_image_device": (C.c_int, [_VP, C.POINTER(_VP)]),
    "mgs_stager_recv_to_host": (C.c_int, [_VP, C.c_int64, C.POINTER(_VP), C.POINTER(C.POINTER(C.c_uint32)),
                                          C.POINTER(C.c_uint64)]),
    "mgs_stager_set_combine": (C.c_int, [_VP, C.c_int]),
    "mgs_synth_u64_records": (C.c_int, [_VP, C.c_uint64, C.c_uint64,
                                        C.c_uint64, C.c_int, _VP]),
    "mgs_unpack_device": (C.c_int, [_VP, _VP, C.c_uint64, _VP, C.c_uint64,
                                    C.c_int64, C.c_int64, C.POINTER(C.c_uint64),
                                    C.POINTER(PageCut), C.c_int,
                                    C.POINTER(C.c_int), _VP]),
    "mgs_combine_device": (C.c_int, [_VP, _VP, _VP, C.c_uint64, C.c_uint64, C.c_int, _VP, _VP,
                                     C.c_int64, _VP, C.POINTER(Round)]),
    "mgs_group_device": (C.c_int, [_VP, _VP, _VP, C.c_uint64, _VP, _VP, C.c_uint64, _VP,
                                   C.c_int64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                   C.c_int, C.POINTER(C.c_int), _VP]),
    "mgs_tokenize_device": (C.c_int, [_VP, _VP, C.c_uint64, C.c_uint64, _VP, C.c_uint64,
                                      _VP, C.c_uint64, C.POINTER(C.c_uint64),
                                      C.POINTER(C.c_uint64), _VP]),
}

_lib = None


def library_path():
    """The in-tree build, or another build of the same ABI named by MGS_LIBRARY
    (for A/B timing of two builds on one box)."""
    return os.environ.get("MGS_LIBRARY") or _build.LIB


def load():
    """Load libmimir_gpu_shuffle.so and type every exported entry point."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build the HIP library first "
            "(python -m mimir_amd.build); there is no CPU fallback")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def geometry(comm_buf_size, nranks):
    """(buf_size, unit_size, type_log_bytes) of the reference's send buffer."""
    bs, us, tl = C.c_int64(), C.c_int(), C.c_int()
    rc = load().mgs_geometry(comm_buf_size, nranks, C.byref(bs), C.byref(us),
                             C.byref(tl))
    if rc != OK:
        raise MgsError(rc, "The communication buffer is too small")
    return bs.value, us.value, tl.value


def hashlittle(data: bytes, initval=0):
    buf = C.create_string_buffer(bytes(data), len(data))
    return load().mgs_hashlittle_host(buf, len(data), initval)


def dest_of_hash(h, nranks):
    return load().mgs_dest_of_hash_host(h, nranks)
