"""Test doubles built on the CPU oracle (test infrastructure only).

simulate_shuffle  the whole multi-rank map-side shuffle as the reference runs
                  it (write -> rounds -> exchange_kv -> save_data -> wait),
                  single process, oracle only: what every rank's sink must
                  receive, in order.
OraclePacker      stands in for mimir_amd.packer.DevicePacker on CPU tensors,
                  so the host logic (staging, rounds, exchange, termination
                  vote) can be exercised over gloo without a GPU.
"""
import ctypes as C

import numpy as np

from oracle import oracle as O


def olayout(lay):
    return O.Layout(lay.key_kind, lay.key_bytes, lay.key_count, lay.val_kind,
                    lay.val_bytes, lay.val_count)


def split_records(orc, olay, buf):
    offs = orc.parse_records(olay, buf)
    b = buf.tobytes()
    return [b[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]


def simulate_shuffle(orc, lay, streams, P, buf_size, combine_op=None):
    """streams[r] = (blob, rec_off or None, n).  Returns (sinks, stats):
    sinks[r] = list of wire-form records in the order rank r's `out` sees
    them; stats[r] = dict(kvcount, shuffle_times, send_bytes, recv_bytes).
    With combine_op the rounds are those of CombineCollectiveShuffler."""
    olay = olayout(lay)
    rounds = []  # rounds[r] = list of oracle round dicts
    for r in range(P):
        blob, offs, n = streams[r]
        rs, start = [], 0
        while True:
            if combine_op is None:
                rd = orc.pack_round(olay, blob, offs, n, start, P, r, buf_size)
            else:
                rd = orc.combine_round(olay, blob, offs, n, start, P, r, buf_size, combine_op)
            rs.append(rd)
            start = rd["next"]
            if start >= n:
                break
        rounds.append(rs)
    nex = max(len(rs) for rs in rounds)  # data exchanges (>= 1 each rank)
    sinks = [[] for _ in range(P)]
    stats = [dict(kvcount=sum(rd["packed_records"] for rd in rounds[r]),
                  shuffle_times=nex, send_bytes=0, recv_bytes=0) for r in range(P)]
    for j in range(nex):
        for r in range(P):
            if j < len(rounds[r]):
                sinks[r].extend(split_records(orc, olay, rounds[r][j]["local"]))
                stats[r]["send_bytes"] += int(rounds[r][j]["send_offset"].sum())
        for r in range(P):
            for src in range(P):
                if j < len(rounds[src]):
                    part = rounds[src][j]["partitions"][r]
                    stats[r]["recv_bytes"] += len(part)
                    sinks[r].extend(split_records(orc, olay, part))
    return sinks, stats


def simulate_streams(orc, lay, streams, P, buf_size):
    """Per-source view of the same shuffle, for shufflers whose exchanges
    complete asynchronously: per[r][src] = the records rank r receives from
    rank src (src == r: its own local-bypass records), in order; rounds[r] =
    the send_offset vector of each of rank r's rounds that carries data."""
    olay = olayout(lay)
    per = [[[] for _ in range(P)] for _ in range(P)]
    rounds = [[] for _ in range(P)]
    kvcount = [0] * P
    for src in range(P):
        blob, offs, n = streams[src]
        start = 0
        while True:
            rd = orc.pack_round(olay, blob, offs, n, start, P, src, buf_size)
            per[src][src].extend(split_records(orc, olay, rd["local"]))
            for r in range(P):
                if r != src:
                    per[r][src].extend(split_records(orc, olay, rd["partitions"][r]))
            if int(rd["send_offset"].sum()) > 0:
                rounds[src].append(rd["send_offset"].tolist())
            kvcount[src] += rd["packed_records"]
            start = rd["next"]
            if start >= n:
                break
    return per, rounds, kvcount


class _Round:
    pass


class OraclePacker:
    def __init__(self, layout, nranks, rank, buf_size, max_records):
        self.orc = O.Oracle()
        self.olay = olayout(layout)
        self.P, self.rank, self.buf_size = nranks, rank, buf_size

    def pack(self, blob, n, send_image, start=0, rec_off=None, init_offset=None,
             local_out=None, **_):
        b = blob.numpy()
        offs = None
        if rec_off is not None:
            offs = np.ascontiguousarray(rec_off.numpy().view(np.uint32).astype(np.uint64))
        so = np.zeros(self.P, dtype=np.int64)
        if init_offset is not None:
            so[:] = np.asarray(init_offset, dtype=np.int64)[:self.P]
        img = send_image.numpy()
        loc = local_out.numpy()
        lb, lr, pr = C.c_int64(), C.c_uint64(), C.c_uint64()
        nxt = self.orc.L.mo_pack_more(
            C.byref(self.olay), b.ctypes.data, offs.ctypes.data if offs is not None else None,
            self.olay.fixed_len() or 0, n, start, self.P, self.rank, self.buf_size,
            img.ctypes.data, so.ctypes.data, loc.ctypes.data, len(loc),
            C.byref(lb), C.byref(lr), C.byref(pr))
        if nxt == O.U64_MAX:
            raise ValueError("oracle packer: bad record")
        r = _Round()
        r.next, r.send_offset = nxt, so
        r.packed_records, r.local_records, r.local_bytes = pr.value, lr.value, lb.value
        return r

    def combine(self, blob, n, send_image, start=0, rec_off=None, local_out=None, op=0):
        b = blob.numpy()
        offs = None
        if rec_off is not None:
            offs = np.ascontiguousarray(rec_off.numpy().view(np.uint32).astype(np.uint64))
        so = np.zeros(self.P, dtype=np.int64)
        img = send_image.numpy()
        loc = local_out.numpy()
        lb, lr, pr = C.c_int64(), C.c_uint64(), C.c_uint64()
        nxt = self.orc.L.mo_combine_round(
            C.byref(self.olay), b.ctypes.data, offs.ctypes.data if offs is not None else None,
            self.olay.fixed_len() or 0, n, start, self.P, self.rank, self.buf_size, op,
            img.ctypes.data, so.ctypes.data, loc.ctypes.data, len(loc),
            C.byref(lb), C.byref(lr), C.byref(pr))
        if nxt == O.U64_MAX:
            raise ValueError("oracle packer: bad record")
        r = _Round()
        r.next, r.send_offset = nxt, so
        r.packed_records, r.local_records, r.local_bytes = pr.value, lr.value, lb.value
        return r

    def close(self):
        pass
