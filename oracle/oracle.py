"""ctypes front-end to the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product path (mimir_amd/) never does.

`Oracle` wraps oracle/libmimir_oracle.so (the plain-C restatement in
mimir_oracle.c).  `Reference` wraps oracle/_ref/libmimir_ref.so, which is the
reference's own hash.cpp and serializer.h bytestream templates compiled from
/root/reference by oracle/Makefile; it exists only where that build ran.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "libmimir_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libmimir_ref.so")
REF_SER_SO = os.path.join(_HERE, "_ref", "libmimir_ref_ser.so")

KEY_FIXED, KEY_CSTR = 0, 1
VAL_VOID, VAL_FIXED, VAL_CSTR = 0, 1, 2
U64_MAX = (1 << 64) - 1


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
        """Record size when both halves are fixed width, else None."""
        if self.key_kind == KEY_FIXED and self.val_kind != VAL_CSTR:
            return self.key_bytes + (self.val_bytes
                                     if self.val_kind == VAL_FIXED else 0)
        return None


def build(force=False):
    """Compile libmimir_oracle.so (and _ref/ when /root/reference exists)."""
    if force or not os.path.exists(ORACLE_SO) or (
            os.path.getmtime(ORACLE_SO)
            < os.path.getmtime(os.path.join(_HERE, "mimir_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "libmimir_oracle.so"],
                              stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/src/hash.cpp") and (
            force or not os.path.exists(REF_SO)):
        subprocess.check_call(["make", "-C", _HERE, "ref"],
                              stdout=subprocess.DEVNULL)


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8)) if a is not None else None


class Oracle:
    def __init__(self):
        build()
        L = C.CDLL(ORACLE_SO)
        L.mo_hashlittle.restype = C.c_uint32
        L.mo_hashlittle.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
        L.mo_geometry.restype = C.c_int
        L.mo_geometry.argtypes = [C.c_int64, C.c_int, C.POINTER(C.c_int64),
                                  C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.mo_key_len.restype = C.c_int64
        L.mo_key_len.argtypes = [C.POINTER(Layout), C.c_void_p, C.c_int64]
        L.mo_dest.restype = C.c_int
        L.mo_dest.argtypes = [C.POINTER(Layout), C.c_void_p, C.c_void_p,
                              C.c_int64, C.c_uint64, C.c_uint32, C.c_void_p,
                              C.c_void_p]
        L.mo_pack_round.restype = C.c_uint64
        L.mo_pack_round.argtypes = [
            C.POINTER(Layout), C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64,
            C.c_uint64, C.c_uint32, C.c_int32, C.c_int64, C.c_void_p,
            C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64),
            C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.mo_pack_more.restype = C.c_uint64
        L.mo_pack_more.argtypes = L.mo_pack_round.argtypes
        L.mo_parse_records.restype = C.c_int64
        L.mo_parse_records.argtypes = [C.POINTER(Layout), C.c_void_p,
                                       C.c_int64, C.c_void_p, C.c_int64]
        L.mo_page_cuts.restype = C.c_int64
        L.mo_page_cuts.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                   C.c_void_p, C.c_int64]
        L.mo_combine_round.restype = C.c_uint64
        L.mo_combine_round.argtypes = [
            C.POINTER(Layout), C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64,
            C.c_uint64, C.c_uint32, C.c_int32, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
            C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_uint64),
            C.POINTER(C.c_uint64)]
        L.mo_kmv_convert.restype = C.c_int64
        L.mo_kmv_convert.argtypes = [C.POINTER(Layout), C.c_void_p, C.c_void_p, C.c_int64,
                                     C.c_uint64, C.c_uint32, C.c_int64, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.mo_tokenize.restype = C.c_int64
        L.mo_tokenize.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_void_p,
                                  C.c_int64, C.c_void_p, C.c_int64]
        self.L = L

    def combine_round(self, lay, blob, rec_off, n, start, nranks, self_rank, buf_size, op=0):
        """One round of CombineCollectiveShuffler::write; same result dict as pack_round."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offs = self._offs(rec_off)
        img = np.zeros(nranks * buf_size, dtype=np.uint8)
        so = np.zeros(nranks, dtype=np.int64)
        loc = np.zeros(max(len(blob), 1), dtype=np.uint8)
        lb, lr, pr = C.c_int64(), C.c_uint64(), C.c_uint64()
        nxt = self.L.mo_combine_round(
            C.byref(lay), blob.ctypes.data, offs.ctypes.data if offs is not None else None,
            lay.fixed_len() if offs is None else 0, n, start, nranks, self_rank, buf_size, op,
            img.ctypes.data, so.ctypes.data, loc.ctypes.data, len(loc),
            C.byref(lb), C.byref(lr), C.byref(pr))
        if nxt == 2**64 - 1:
            raise ValueError("oracle: combine_round failed")
        return {"next": int(nxt), "send_offset": so, "local_bytes": lb.value,
                "local_records": lr.value, "packed_records": pr.value,
                "partitions": [img[d * buf_size:d * buf_size + so[d]] for d in range(nranks)],
                "local": loc[:lb.value]}

    def kmv_convert(self, lay, blob, rec_off, n, hashscale=1, page_bytes=64 << 20):
        """KMVContainer::convert: (group_first u32[G], group_start u64[G+1],
        values uint8, cut groups u64[])."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offs = self._offs(rec_off)
        vb = lay.val_bytes if lay.val_kind == VAL_FIXED else 0
        first = np.zeros(max(n, 1), dtype=np.uint32)
        start = np.zeros(n + 1, dtype=np.uint64)
        values = np.zeros(max(n * vb, 1), dtype=np.uint8)
        cuts = np.zeros(4096, dtype=np.uint64)
        nc = C.c_int64()
        G = self.L.mo_kmv_convert(C.byref(lay), blob.ctypes.data,
                                  offs.ctypes.data if offs is not None else None,
                                  lay.fixed_len() if offs is None else 0, n, hashscale, page_bytes,
                                  first.ctypes.data, start.ctypes.data, values.ctypes.data,
                                  cuts.ctypes.data, len(cuts), C.byref(nc))
        if G < 0:
            raise ValueError("oracle: kmv_convert failed")
        return first[:G], start[:G + 1], values[:n * vb], cuts[:nc.value]

    def tokenize(self, text, value=1):
        """wordcount's map over a text: (records uint8, rec_off uint64[n+1])."""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        cap = 5 * len(text) + 16
        out = np.zeros(cap, dtype=np.uint8)
        offs = np.zeros(len(text) // 2 + 2, dtype=np.uint64)
        n = self.L.mo_tokenize(text.ctypes.data, len(text), value, out.ctypes.data,
                               cap, offs.ctypes.data, len(offs) - 1)
        if n < 0:
            raise ValueError("oracle: tokenize output does not fit")
        return out[:int(offs[n])], offs[:n + 1]

    def page_cuts(self, rec_off, page_bytes, first_fill=0):
        """Indices of the records that open a new container page."""
        rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
        n = len(rec_off) - 1
        out = np.zeros(max(n, 1), dtype=np.uint64)
        k = self.L.mo_page_cuts(rec_off.ctypes.data, n, page_bytes, first_fill,
                                out.ctypes.data, len(out))
        if k < 0:
            raise ValueError("oracle: record larger than a page")
        return out[:k]

    def hashlittle(self, data: bytes, initval=0):
        buf = C.create_string_buffer(bytes(data), len(data))
        return self.L.mo_hashlittle(buf, len(data), initval)

    def geometry(self, comm_buf_size, nranks):
        bs, us, tl = C.c_int64(), C.c_int(), C.c_int()
        rc = self.L.mo_geometry(comm_buf_size, nranks, C.byref(bs),
                                C.byref(us), C.byref(tl))
        if rc != 0:
            raise ValueError("communication buffer is too small")
        return bs.value, us.value, tl.value

    @staticmethod
    def _offs(rec_off):
        if rec_off is None:
            return None
        return np.ascontiguousarray(rec_off, dtype=np.uint64)

    def dest(self, lay, blob, rec_off, n, nranks):
        """-> (hash uint32[n], dest int32[n])"""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offs = self._offs(rec_off)
        h = np.empty(n, dtype=np.uint32)
        d = np.empty(n, dtype=np.int32)
        rc = self.L.mo_dest(C.byref(lay), blob.ctypes.data,
                            offs.ctypes.data if offs is not None else None,
                            lay.fixed_len() or 0, n, nranks, h.ctypes.data,
                            d.ctypes.data)
        if rc != 0:
            raise ValueError("malformed record")
        return h, d

    def pack_round(self, lay, blob, rec_off, n, start, nranks, self_rank,
                   buf_size, want_bytes=True, init_offset=None):
        """One CollectiveShuffler::write round.  Returns a dict with next,
        send_offset[int64 P], partitions (list of bytes-like np arrays),
        local (np uint8), local_records, packed_records."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offs = self._offs(rec_off)
        so = np.zeros(nranks, dtype=np.int64)
        img = np.zeros(nranks * buf_size, dtype=np.uint8) if want_bytes else None
        total = int(offs[n] - offs[start]) if offs is not None else (
            (n - start) * lay.fixed_len())
        loc = np.zeros(max(total, 1), dtype=np.uint8) if want_bytes else None
        lb, lr, pr = C.c_int64(), C.c_uint64(), C.c_uint64()
        fn = self.L.mo_pack_round
        if init_offset is not None:  # the round goes on from partly filled partitions
            so[:] = np.asarray(init_offset, dtype=np.int64)[:nranks]
            if self_rank >= 0:
                so[self_rank] = 0
            fn = self.L.mo_pack_more
        nxt = fn(
            C.byref(lay), blob.ctypes.data,
            offs.ctypes.data if offs is not None else None,
            lay.fixed_len() or 0, n, start, nranks, self_rank, buf_size,
            img.ctypes.data if img is not None else None, so.ctypes.data,
            loc.ctypes.data if loc is not None else None, max(total, 1),
            C.byref(lb), C.byref(lr), C.byref(pr))
        if nxt == U64_MAX:
            raise ValueError("oracle: malformed / oversize record")
        out = {"next": nxt, "send_offset": so, "local_bytes": lb.value,
               "local_records": lr.value, "packed_records": pr.value}
        if want_bytes:
            out["image"] = img
            out["partitions"] = [img[d * buf_size:d * buf_size + so[d]]
                                 for d in range(nranks)]
            out["local"] = loc[:lb.value]
        return out

    def parse_records(self, lay, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        offs = np.zeros(len(buf) + 1, dtype=np.uint64)
        n = self.L.mo_parse_records(C.byref(lay), buf.ctypes.data, len(buf),
                                    offs.ctypes.data, len(buf))
        if n < 0:
            raise ValueError("oracle: malformed receive image")
        return offs[:n + 1]


class Reference:
    """The reference's own compiled hash.cpp / serializer.h (oracle/_ref)."""

    def __init__(self):
        build()
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO)
        L = C.CDLL(REF_SO)
        L.ref_hashlittle.restype = C.c_uint32
        L.ref_hashlittle.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
        L.ref_kv_cstr_u64.restype = C.c_int
        L.ref_kv_cstr_u64.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p,
                                      C.c_int]
        L.ref_kv_u64_u64.restype = C.c_int
        L.ref_kv_u64_u64.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p,
                                     C.c_int]
        L.ref_kv_chars_u64.restype = C.c_int
        L.ref_kv_chars_u64.argtypes = [C.c_char_p, C.c_int, C.c_uint64,
                                       C.c_void_p, C.c_int]
        L.ref_kv_cstr_blob16.restype = C.c_int
        L.ref_kv_cstr_blob16.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p,
                                         C.c_int]
        L.ref_key_size_cstr.restype = C.c_int
        L.ref_key_size_cstr.argtypes = [C.c_char_p]
        L.ref_wordcount_map.restype = C.c_long
        L.ref_wordcount_map.argtypes = [C.c_void_p, C.c_long, C.c_uint64, C.c_void_p,
                                        C.c_long, C.c_void_p, C.c_long]
        for name, extra in (("ref_parse_u64_u64", []), ("ref_parse_chars_u64", [C.c_int]),
                            ("ref_parse_cstr_fixed", [C.c_int])):
            fn = getattr(L, name)
            fn.restype = C.c_long
            fn.argtypes = [C.c_void_p, C.c_long] + extra + [C.c_void_p, C.c_long]
        self.L = L

    def parse_records(self, shape, buf, arg=0):
        """Record offsets of a receive stream, found by the reference's own
        from_bytes / kv_from_bytes walk (save_data).  shape: "u64_u64",
        "chars_u64" (arg = keycount) or "cstr_fixed" (arg = value bytes)."""
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        work = np.zeros(len(buf) + 8, dtype=np.uint8)  # strlen may look one byte past a bad stream
        work[:len(buf)] = buf
        offs = np.zeros(len(buf) + 2, dtype=np.uint64)
        if shape == "u64_u64":
            n = self.L.ref_parse_u64_u64(work.ctypes.data, len(buf), offs.ctypes.data, len(buf) + 1)
        elif shape == "chars_u64":
            n = self.L.ref_parse_chars_u64(work.ctypes.data, len(buf), arg, offs.ctypes.data, len(buf) + 1)
        else:
            n = self.L.ref_parse_cstr_fixed(work.ctypes.data, len(buf), arg, offs.ctypes.data, len(buf) + 1)
        if n < 0:
            raise ValueError(f"reference parse failed ({n})")
        return offs[:n + 1]

    def wordcount_map(self, text, value=1):
        """The reference's own line framing + serialisation over a text."""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        work = np.zeros(len(text) + 8, dtype=np.uint8)  # modified in place, one spare byte
        work[:len(text)] = text
        cap = 5 * len(text) + 16
        out = np.zeros(cap, dtype=np.uint8)
        offs = np.zeros(len(text) // 2 + 2, dtype=np.uint64)
        n = self.L.ref_wordcount_map(work.ctypes.data, len(text), value, out.ctypes.data,
                                     cap, offs.ctypes.data, len(offs) - 1)
        if n < 0:
            raise ValueError("reference: output does not fit")
        return out[:int(offs[n])], offs[:n + 1]

    def hashlittle(self, data: bytes, initval=0, misalign=0):
        """Hash `data` placed `misalign` bytes past an 8-aligned address, so
        all three load strategies of src/hash.cpp:166/:286/:344 get used."""
        raw = C.create_string_buffer(len(data) + 16)
        base = C.addressof(raw)
        start = (-base) % 8 + misalign
        C.memmove(base + start, bytes(data), len(data))
        return self.L.ref_hashlittle(base + start, len(data), initval)

    def _call(self, fn, *args, cap=4096):
        buf = C.create_string_buffer(cap)
        n = fn(*args, buf, cap)
        return None if n < 0 else buf.raw[:n]

    def kv_cstr_u64(self, key: bytes, val: int, cap=4096):
        return self._call(self.L.ref_kv_cstr_u64, key, val, cap=cap)

    def kv_u64_u64(self, key: int, val: int, cap=4096):
        return self._call(self.L.ref_kv_u64_u64, key, val, cap=cap)

    def kv_chars_u64(self, key: bytes, keycount: int, val: int, cap=4096):
        return self._call(self.L.ref_kv_chars_u64, key, keycount, val, cap=cap)

    def kv_cstr_blob16(self, key: bytes, val16: bytes, cap=4096):
        v = C.create_string_buffer(bytes(val16), 16)
        return self._call(self.L.ref_kv_cstr_blob16, key, v, cap=cap)


class ReferenceSerializer:
    """The reference's own Serializer<K,V>::kv_to_bytes / get_hash_code
    (oracle/_ref/libmimir_ref_ser.so, built from oracle/ref_probe_ser.cpp)."""

    def __init__(self):
        build()
        if not os.path.exists(REF_SER_SO):
            raise FileNotFoundError(REF_SER_SO)
        L = C.CDLL(REF_SER_SO)
        L.ser_kv_cstr_u64.restype = C.c_int
        L.ser_kv_cstr_u64.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_uint64, C.c_void_p, C.c_int]
        L.ser_hash_cstr.restype = C.c_uint32
        L.ser_hash_cstr.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.ser_kv_u64_u64.restype = C.c_int
        L.ser_kv_u64_u64.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_int]
        L.ser_hash_u64.restype = C.c_uint32
        L.ser_hash_u64.argtypes = [C.c_uint64]
        L.ser_kv_chars_u64.restype = C.c_int
        L.ser_kv_chars_u64.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.c_void_p, C.c_int]
        L.ser_hash_chars.restype = C.c_uint32
        L.ser_hash_chars.argtypes = [C.c_char_p, C.c_int]
        L.ser_kv_cstr_blob16.restype = C.c_int
        L.ser_kv_cstr_blob16.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int]
        L.ser_write_loop_u64_u64.restype = C.c_long
        L.ser_write_loop_u64_u64.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_long, C.c_void_p]
        self.L = L

    def write_loop_u64_u64(self, pairs, n, nranks, send_buffer, buf_size, send_offset):
        """CollectiveShuffler::write's per-record work (src/collectiveshuffler.h:131-166) over
        n <u64,u64> pairs with the reference's compiled get_hash_code / hashlittle /
        kv_to_bytes: numpy arrays in, pairs packed out (stops at the first that does not fit)."""
        return self.L.ser_write_loop_u64_u64(pairs.ctypes.data, n, nranks, send_buffer.ctypes.data, buf_size,
                                             send_offset.ctypes.data)

    @staticmethod
    def _run(fn, args, bufsize, room=4096):
        """kv_to_bytes into a buffer of `room` bytes (filled with 0xEE) that the call is told
        holds `bufsize`: -> (return value, the bytes it wrote = the prefix that changed)."""
        buf = C.create_string_buffer(b"\xee" * room, room)
        rc = fn(*args, buf, bufsize)
        raw = buf.raw
        written = room
        while written > 0 and raw[written - 1] == 0xEE:
            written -= 1
        return rc, raw[:written]

    def kv_to_bytes(self, shape, key, val, bufsize, keycount=1):
        """shape: "cstr_u64" (key: bytes, or a pair of bytes with keycount 2), "u64_u64",
        "chars_u64" (keycount characters of key), "cstr_blob16" (val: 16 bytes)."""
        if shape == "cstr_u64":
            k0, k1 = (key, None) if keycount == 1 else key
            return self._run(self.L.ser_kv_cstr_u64, (k0, k1, keycount, val), bufsize)
        if shape == "u64_u64":
            return self._run(self.L.ser_kv_u64_u64, (key, val), bufsize)
        if shape == "chars_u64":
            return self._run(self.L.ser_kv_chars_u64, (key, keycount, val), bufsize)
        if shape == "cstr_blob16":
            v = C.create_string_buffer(bytes(val), 16)
            return self._run(self.L.ser_kv_cstr_blob16, (key, v), bufsize)
        raise ValueError(shape)

    def hash_code(self, shape, key, keycount=1):
        if shape in ("cstr_u64", "cstr_blob16"):
            k0, k1 = (key, None) if keycount == 1 else key
            return self.L.ser_hash_cstr(k0, k1, keycount)
        if shape == "u64_u64":
            return self.L.ser_hash_u64(key)
        if shape == "chars_u64":
            return self.L.ser_hash_chars(key, keycount)
        raise ValueError(shape)
