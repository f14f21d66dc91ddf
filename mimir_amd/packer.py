"""Device-resident packer: torch tensors in, send image out, through the C ABI.

torch is plumbing here (device memory, streams); all arithmetic happens in the
HIP kernels behind include/mimir_gpu_shuffle.h.
"""
import ctypes as C

import numpy as np
import torch

from . import capi


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class RoundResult:
    """Host copy of one mgs_round."""

    def __init__(self, r, nranks):
        self.next = int(r.next)
        self.packed_records = int(r.packed_records)
        self.local_records = int(r.local_records)
        self.local_bytes = int(r.local_bytes)
        self.send_offset = np.array(r.send_offset[:nranks], dtype=np.int64)

    def __repr__(self):
        return (f"RoundResult(next={self.next}, packed={self.packed_records}, "
                f"local={self.local_records}/{self.local_bytes}B, "
                f"send_offset={self.send_offset.tolist()})")


class RoundsResult:
    """Host copy of a many-round pack call: totals plus one row per round."""

    def __init__(self, r, nxt, so, lrec, lbytes, nranks, base=None):
        n = int(r.n_rounds)
        # streams layout: byte of each destination stream at which the round's partition begins
        self.round_base = None if base is None else base[:n * nranks].reshape(n, nranks).copy()
        self.n_rounds = n
        self.next = int(r.next)
        self.packed_records = int(r.packed_records)
        self.local_records = int(r.local_records)
        self.local_bytes = int(r.local_bytes)
        self.round_next = nxt[:n].copy()
        self.round_send_offset = so[:n * nranks].reshape(n, nranks).copy()
        self.round_local_records = lrec[:n].copy()
        self.round_local_bytes = lbytes[:n].copy()


def merge_stream_rounds(a, b):
    """The rounds of two consecutive streams-layout calls on the same destination streams
    (the second started with stream_fill / open_round_base taken from the first) as one
    table: the first round of b is the open round of a going on."""
    if a is None:
        return b
    assert np.array_equal(a.round_base[-1], b.round_base[0]), "the second call did not go on in the open round"
    m = RoundsResult.__new__(RoundsResult)
    m.n_rounds = a.n_rounds + b.n_rounds - 1
    m.next = b.next
    m.packed_records = a.packed_records + b.packed_records
    m.local_records = a.local_records + b.local_records
    m.local_bytes = a.local_bytes + b.local_bytes
    m.round_next = np.concatenate([a.round_next[:-1], b.round_next])
    m.round_send_offset = np.concatenate([a.round_send_offset[:-1], b.round_send_offset])
    m.round_base = np.concatenate([a.round_base[:-1], b.round_base])
    lr, lb = b.round_local_records.copy(), b.round_local_bytes.copy()
    lr[0] += a.round_local_records[-1]
    lb[0] += a.round_local_bytes[-1]
    m.round_local_records = np.concatenate([a.round_local_records[:-1], lr])
    m.round_local_bytes = np.concatenate([a.round_local_bytes[:-1], lb])
    return m


class DevicePacker:
    """One rank's pack state (mgs_handle) on one GPU.

    layout   capi.Layout
    nranks   number of destinations P
    rank     this rank, or -1 to pack all P partitions (no local bypass)
    buf_size bytes per destination partition (reference geometry:
             capi.geometry(COMM_BUF_SIZE, nranks)[0])
    """

    def __init__(self, layout, nranks, rank, buf_size, max_records,
                 device=0, flags=0):
        self.lib = capi.load()
        self.layout = layout
        self.nranks = nranks
        self.rank = rank
        self.buf_size = int(buf_size)
        self.device = int(device)
        self.max_records = int(max_records)
        desc = capi.Desc(nranks, rank, layout, self.buf_size, self.device, int(flags),
                         self.max_records)
        h = C.c_void_p()
        rc = self.lib.mgs_create(C.byref(desc), C.byref(h))
        if rc != capi.OK:
            raise capi.MgsError(rc, self.lib.mgs_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.mgs_destroy(self.h)
            self.h = None

    __del__ = close

    def _check(self, rc):
        if rc != capi.OK:
            raise capi.MgsError(rc, self.lib.mgs_last_error(self.h).decode())

    def last_algo(self):
        return self.lib.mgs_last_algo(self.h).decode()

    def last_kernel(self):
        """Name of the dominant kernel of the last pack call."""
        return self.lib.mgs_last_kernel(self.h).decode()

    def last_kernel_ms(self):
        """Device time of the dominant kernel of the last completed call."""
        ms = C.c_float()
        self._check(self.lib.mgs_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    def tuning_name(self):
        return self.lib.mgs_tuning_name(self.lib.mgs_get_tuning(self.h)).decode()

    def set_tuning(self, variant):
        self._check(self.lib.mgs_set_tuning(self.h, variant))

    def debug_stats(self):
        out = (C.c_uint64 * 8)()
        self._check(self.lib.mgs_debug_stats(self.h, out))
        return list(out)

    def finder_stats(self):
        out = (C.c_uint64 * 4)()
        self._check(self.lib.mgs_finder_stats(self.h, out))
        return list(out)

    def copy_calibration(self, src, dst, n_records):
        """ms to stream n_records 16-byte records src -> dst (tile loop only)."""
        ms = C.c_float()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.mgs_copy_calibration(
            self.h, _ptr(src), _ptr(dst), n_records, C.c_void_p(stream), C.byref(ms)))
        return ms.value

    def new_send_image(self):
        return torch.empty(self.nranks * self.buf_size, dtype=torch.uint8,
                           device=f"cuda:{self.device}")

    def _args(self, records, rec_off, n_end, start, init_offset, send_image,
              local_out, dest_out, algo, stream):
        if init_offset is not None:
            init = np.ascontiguousarray(init_offset, dtype=np.int64)
            assert init.shape[0] >= self.nranks
            init_p = init.ctypes.data_as(C.c_void_p)
        else:
            init, init_p = None, None
        if stream is None:
            # torch's default stream is the legacy null stream: the library then uses a
            # blocking stream of its own, ordered after what torch has enqueued there
            stream = torch.cuda.current_stream(self.device).cuda_stream
        local_cap = local_out.numel() if local_out is not None else 0
        return (self.h, _ptr(records), _ptr(rec_off), n_end, start, init_p,
                _ptr(send_image), _ptr(local_out), local_cap, _ptr(dest_out),
                algo, C.c_void_p(stream)), init

    def pack(self, records, n_end, send_image, start=0, rec_off=None,
             init_offset=None, local_out=None, dest_out=None,
             algo=capi.ALGO_AUTO, stream=None):
        """Blocking CollectiveShuffler::write over records [start, n_end)."""
        args, _keep = self._args(records, rec_off, n_end, start, init_offset,
                                 send_image, local_out, dest_out, algo, stream)
        r = capi.Round()
        self._check(self.lib.mgs_pack_device(*args, C.byref(r)))
        return RoundResult(r, self.nranks)

    def _rounds_buffers(self, max_rounds):
        return (np.zeros(max_rounds, dtype=np.uint64), np.zeros(max_rounds * self.nranks, dtype=np.int64),
                np.zeros(max_rounds, dtype=np.uint64), np.zeros(max_rounds, dtype=np.int64))

    def pack_rounds(self, records, n_end, send_images, image_stride, max_rounds, start=0, rec_off=None,
                    init_offset=None, local_out=None, dest_out=None, algo=capi.ALGO_AUTO, stream=None):
        """mgs_pack_rounds_device: records [start, n_end) into up to max_rounds send images
        (image k at send_images[k * image_stride:]), round boundaries as the reference cuts them."""
        args, _keep = self._args(records, rec_off, n_end, start, init_offset, send_images, local_out,
                                 dest_out, algo, stream)
        h, rec, off, n_end, start, init_p, img, loc, lcap, dst, algo, st = args
        nxt, so, lrec, lbytes = self._rounds_buffers(max_rounds)
        r = capi.Rounds()
        self._check(self.lib.mgs_pack_rounds_device(
            h, rec, off, n_end, start, init_p, img, image_stride, max_rounds, loc, lcap, dst, algo, st,
            C.byref(r), nxt.ctypes.data_as(C.c_void_p), so.ctypes.data_as(C.c_void_p),
            lrec.ctypes.data_as(C.c_void_p), lbytes.ctypes.data_as(C.c_void_p)))
        return RoundsResult(r, nxt, so, lrec, lbytes, self.nranks)

    def pack_rounds_async(self, records, n_end, send_images, image_stride, max_rounds, start=0, rec_off=None,
                          init_offset=None, local_out=None, dest_out=None, algo=capi.ALGO_AUTO, stream=None):
        args, _keep = self._args(records, rec_off, n_end, start, init_offset, send_images, local_out,
                                 dest_out, algo, stream)
        h, rec, off, n_end, start, init_p, img, loc, lcap, dst, algo, st = args
        self._pending_rounds = max_rounds
        self._check(self.lib.mgs_pack_rounds_device_async(
            h, rec, off, n_end, start, init_p, img, image_stride, max_rounds, loc, lcap, dst, algo, st))

    def rounds_result(self):
        nxt, so, lrec, lbytes = self._rounds_buffers(self._pending_rounds)
        r = capi.Rounds()
        self._check(self.lib.mgs_pack_rounds_result(
            self.h, C.byref(r), nxt.ctypes.data_as(C.c_void_p), so.ctypes.data_as(C.c_void_p),
            lrec.ctypes.data_as(C.c_void_p), lbytes.ctypes.data_as(C.c_void_p)))
        return RoundsResult(r, nxt, so, lrec, lbytes, self.nranks)

    def _streams_args(self, records, rec_off, n_end, start, stream_fill, open_round_base, streams, stream_cap,
                      max_rounds, local_out, dest_out, stream):
        keep = []

        def host(a):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.int64)
            assert a.shape[0] >= self.nranks
            keep.append(a)
            return a.ctypes.data_as(C.c_void_p)
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        local_cap = local_out.numel() if local_out is not None else 0
        return (self.h, _ptr(records), _ptr(rec_off), n_end, start, host(stream_fill), host(open_round_base),
                _ptr(streams), int(stream_cap), max_rounds, _ptr(local_out), local_cap, _ptr(dest_out),
                C.c_void_p(stream)), keep

    def pack_streams(self, records, n_end, streams, stream_cap, max_rounds, start=0, rec_off=None,
                     stream_fill=None, open_round_base=None, local_out=None, dest_out=None, stream=None):
        """mgs_pack_streams_device: records [start, n_end) appended to nranks destination
        streams (stream b at streams[b * stream_cap:]); the reference's rounds come back as
        views: round k's partition b is stream b's bytes [round_base[k, b], + round_send_offset[k, b])."""
        args, _keep = self._streams_args(records, rec_off, n_end, start, stream_fill, open_round_base, streams,
                                         stream_cap, max_rounds, local_out, dest_out, stream)
        nxt, so, lrec, lbytes = self._rounds_buffers(max_rounds)
        base = np.zeros(max_rounds * self.nranks, dtype=np.int64)
        r = capi.Rounds()
        self._check(self.lib.mgs_pack_streams_device(
            *args, C.byref(r), nxt.ctypes.data_as(C.c_void_p), so.ctypes.data_as(C.c_void_p),
            base.ctypes.data_as(C.c_void_p), lrec.ctypes.data_as(C.c_void_p), lbytes.ctypes.data_as(C.c_void_p)))
        return RoundsResult(r, nxt, so, lrec, lbytes, self.nranks, base)

    def pack_streams_async(self, records, n_end, streams, stream_cap, max_rounds, start=0, rec_off=None,
                           stream_fill=None, open_round_base=None, local_out=None, dest_out=None, stream=None):
        args, _keep = self._streams_args(records, rec_off, n_end, start, stream_fill, open_round_base, streams,
                                         stream_cap, max_rounds, local_out, dest_out, stream)
        self._pending_rounds = max_rounds
        self._check(self.lib.mgs_pack_streams_device_async(*args))

    def streams_result(self):
        nxt, so, lrec, lbytes = self._rounds_buffers(self._pending_rounds)
        base = np.zeros(self._pending_rounds * self.nranks, dtype=np.int64)
        r = capi.Rounds()
        self._check(self.lib.mgs_pack_streams_result(
            self.h, C.byref(r), nxt.ctypes.data_as(C.c_void_p), so.ctypes.data_as(C.c_void_p),
            base.ctypes.data_as(C.c_void_p), lrec.ctypes.data_as(C.c_void_p), lbytes.ctypes.data_as(C.c_void_p)))
        return RoundsResult(r, nxt, so, lrec, lbytes, self.nranks, base)

    def pack_noalloc(self, records, n_end, send_image, start=0, rec_off=None,
                     init_offset=None, local_out=None, dest_out=None, stream=None):
        """mgs_pack_device_noalloc: the same call on the library's static device
        scratch, allocating nothing."""
        args, _keep = self._args(records, rec_off, n_end, start, init_offset,
                                 send_image, local_out, dest_out, capi.ALGO_AUTO, stream)
        r = capi.Round()
        self._check(self.lib.mgs_pack_device_noalloc(*args[:10], args[11], C.byref(r)))
        return RoundResult(r, self.nranks)

    def combine(self, records, n_end, send_image, start=0, rec_off=None, local_out=None,
                op=capi.COMBINE_SUM_U64):
        """Blocking CombineCollectiveShuffler::write over records [start, n_end): one
        round, partitions written from their beginning."""
        s = torch.cuda.current_stream(self.device).cuda_stream
        r = capi.Round()
        self._check(self.lib.mgs_combine_device(
            self.h, _ptr(records), _ptr(rec_off), n_end, start, op, _ptr(send_image),
            _ptr(local_out), local_out.numel() if local_out is not None else 0,
            C.c_void_p(s), C.byref(r)))
        return RoundResult(r, self.nranks)

    def pack_async(self, records, n_end, send_image, start=0, rec_off=None,
                   init_offset=None, local_out=None, dest_out=None,
                   algo=capi.ALGO_AUTO, stream=None):
        args, _keep = self._args(records, rec_off, n_end, start, init_offset,
                                 send_image, local_out, dest_out, algo, stream)
        self._check(self.lib.mgs_pack_device_async(*args))

    def result(self):
        r = capi.Round()
        self._check(self.lib.mgs_pack_result(self.h, C.byref(r)))
        return RoundResult(r, self.nranks)


def unpack_stream(packer, stream, nbytes, page_bytes=64 << 20, first_page_fill=0,
                  max_cuts=1024, rec_off=None):
    """Record offsets (device uint32 tensor, n+1 entries) and page cuts of a
    device-resident wire-form stream: save_data + KVContainer::write without
    moving the payload.  Returns (rec_off, n_records, [(first_record, first_byte), ...])."""
    fl = packer.layout.fixed_len()
    cap = (nbytes // fl if fl else nbytes // 2) + 2
    if rec_off is None:
        rec_off = torch.empty(cap, dtype=torch.int32, device=stream.device)
    n = C.c_uint64()
    ncuts = C.c_int()
    cuts = (capi.PageCut * max_cuts)()
    s = torch.cuda.current_stream(packer.device).cuda_stream
    packer._check(packer.lib.mgs_unpack_device(
        packer.h, _ptr(stream), nbytes, _ptr(rec_off), rec_off.numel(), page_bytes, first_page_fill,
        C.byref(n), cuts, max_cuts, C.byref(ncuts), C.c_void_p(s)))
    return rec_off, int(n.value), [(int(cuts[i].first_record), int(cuts[i].first_byte))
                                   for i in range(ncuts.value)]


def group_records(packer, records, n, rec_off=None, page_bytes=64 << 20, max_cuts=1024,
                  group_capacity=None):
    """KV -> KMV on the device (KMVContainer::convert): unique keys in the order of
    first occurrence, values grouped in stream order.  Returns (group_first int32
    tensor [G], group_start int32 tensor [G+1], values uint8 tensor, cuts list)."""
    lay = packer.layout
    vb = lay.val_bytes if lay.val_kind == capi.VAL_FIXED else 0
    cap = n if group_capacity is None else group_capacity
    first = torch.empty(max(cap, 1), dtype=torch.int32, device=records.device)
    start = torch.empty(cap + 1, dtype=torch.int32, device=records.device)
    values = torch.empty(max(n * vb, 1), dtype=torch.uint8, device=records.device)
    G, nc = C.c_uint64(), C.c_int()
    cuts = (C.c_uint64 * max(max_cuts, 1))()
    s = torch.cuda.current_stream(packer.device).cuda_stream
    packer._check(packer.lib.mgs_group_device(
        packer.h, _ptr(records), _ptr(rec_off), n, _ptr(first), _ptr(start), cap, _ptr(values),
        page_bytes, C.byref(G), cuts, max_cuts, C.byref(nc), C.c_void_p(s)))
    g = int(G.value)
    return first[:g], start[:g + 1], values[:n * vb], [int(cuts[i]) for i in range(nc.value)]


def tokenize_text(packer, text, nbytes, value=1, records=None, rec_off=None):
    """wordcount's map on the device: a device-resident text (uint8 tensor, whole
    lines) -> (records uint8 tensor, rec_off int32 tensor, n_records, n_bytes), the
    input form of DevicePacker.pack.  The packer's layout must be <string, 8 bytes>."""
    if records is None:
        records = torch.empty(max(16, nbytes + 9 * (nbytes // 2 + 1)), dtype=torch.uint8, device=text.device)
    if rec_off is None:
        rec_off = torch.empty(nbytes // 2 + 2, dtype=torch.int32, device=text.device)
    n, nb = C.c_uint64(), C.c_uint64()
    s = torch.cuda.current_stream(packer.device).cuda_stream
    packer._check(packer.lib.mgs_tokenize_device(
        packer.h, _ptr(text), nbytes, value, _ptr(records), records.numel(), _ptr(rec_off),
        rec_off.numel(), C.byref(n), C.byref(nb), C.c_void_p(s)))
    return records, rec_off, int(n.value), int(nb.value)


def synth_u64_records(n, seed=42, first=0, device=0, out=None):
    """n <u64,u64> records generated on the device (16*n bytes, uint8)."""
    if out is None:
        out = torch.empty(n * 16, dtype=torch.uint8, device=f"cuda:{device}")
    lib = capi.load()
    stream = torch.cuda.current_stream(device).cuda_stream
    rc = lib.mgs_synth_u64_records(C.c_void_p(out.data_ptr()), n, first, seed,
                                   device, C.c_void_p(stream))
    if rc != capi.OK:
        raise capi.MgsError(rc, "mgs_synth_u64_records failed")
    return out
