"""Host-side shuffler: the reference's CollectiveShuffler interface over the
HIP pack path and a torch.distributed exchange.

`GpuShuffler` mirrors `CollectiveShuffler<K,V>` (src/collectiveshuffler.h):

    open()            allocate buffers (:58-109)
    write(key, val)   route one <key,value> pair (:131-166)
    make_progress()   = exchange_kv() for the blocking shuffler (:168)
    close()           wait(): drain, vote done, exchange until everybody voted
                      (:111-129, :528-561)
    get_record_count() kvcount: records that went through the send buffer
                      (src/baseshuffler.h:265)

What differs is only where the per-record work happens: `write` appends the
pair's wire form to a host staging buffer (the copy the reference makes with
kv_to_bytes, because the caller's pointers are only valid during the call);
`flush` moves the staged batch to the GPU and runs hash -> destination ->
stable pack there.  Round boundaries, local bypass order, the -1 termination
vote and the receive layout are the reference's, bit for bit.

The sink (`out`) receives wire-form records in the order the reference's
`out->write` would see them, as batches: out.write_records(blob, rec_off)
with blob a uint8 numpy array and rec_off the n+1 record offsets.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import capi
from .exchange import exchange_round


def wire_offsets(layout, blob):
    """Record boundaries of back-to-back wire-form records (what save_data
    recovers with kv_from_bytes, src/collectiveshuffler.h:563-585)."""
    n = len(blob)
    fl = layout.fixed_len()
    if fl:
        assert n % fl == 0
        return np.arange(0, n + 1, fl, dtype=np.uint64)
    offs = [0]
    pos = 0
    data = blob.tobytes()
    kc = max(layout.key_count, 1)
    while pos < n:
        if layout.key_kind == capi.KEY_CSTR:
            for _ in range(kc):
                pos = data.index(b"\0", pos) + 1
        else:
            pos += layout.key_bytes
        if layout.val_kind == capi.VAL_FIXED:
            pos += layout.val_bytes
        elif layout.val_kind == capi.VAL_CSTR:
            for _ in range(max(layout.val_count, 1)):
                pos = data.index(b"\0", pos) + 1
        offs.append(pos)
    return np.asarray(offs, dtype=np.uint64)


class ListSink:
    """Minimal Writable: keeps every record it is given, in order."""

    def __init__(self):
        self.records = []

    def write_records(self, blob, rec_off):
        b = blob.tobytes()
        for i in range(len(rec_off) - 1):
            self.records.append(b[int(rec_off[i]):int(rec_off[i + 1])])


class GpuShuffler:
    def __init__(self, out, layout, group=None, comm_buf_size=64 << 20,
                 device=0, stage_bytes=None, packer_factory=None):
        if out is None:
            raise ValueError("Output shuffler cannot be NULL!")  # src/baseshuffler.h:35
        self.out = out
        self.layout = layout
        self.group = group
        self.shuffle_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.shuffle_rank = dist.get_rank(group) if dist.is_initialized() else 0
        # geometry of the constructor, src/collectiveshuffler.h:33-53
        self.buf_size, self.unit_size, self.type_log_bytes = capi.geometry(
            comm_buf_size, self.shuffle_size)
        self.device = device
        self.stage_bytes = int(stage_bytes or comm_buf_size)
        self._packer_factory = packer_factory
        self.kvcount = 0
        self.shuffle_times = 0
        self.done_flag = 0
        self.done_count = 0
        self.send_bytes = 0   # COUNTER_SEND_BYTES, src/collectiveshuffler.h:593
        self.recv_bytes = 0   # COUNTER_RECV_BYTES, :652
        self._open = False

    # ---- BaseObject / Writable surface ------------------------------------
    def open(self):
        P = self.shuffle_size
        fl = self.layout.fixed_len()
        min_rec = fl if fl else 2
        self.max_records = max(self.stage_bytes // min_rec, 1)
        if self._packer_factory is not None:
            self.packer = self._packer_factory(self.layout, P, self.shuffle_rank,
                                               self.buf_size, self.max_records)
            self.dev = torch.device("cpu")
        else:
            from .packer import DevicePacker
            self.packer = DevicePacker(self.layout, P, self.shuffle_rank, self.buf_size,
                                       self.max_records, device=self.device)
            self.dev = torch.device("cuda", self.device)
        backend = dist.get_backend(self.group) if dist.is_initialized() else None
        self.host_exchange = (backend != "nccl") and self.dev.type == "cuda"
        xdev = torch.device("cpu") if (self.host_exchange or self.dev.type == "cpu") else self.dev
        self.send_image = torch.empty(P * self.buf_size, dtype=torch.uint8, device=self.dev)
        self.recv_image = torch.empty(P * self.buf_size, dtype=torch.uint8, device=xdev)
        self.host_send = (torch.empty(P * self.buf_size, dtype=torch.uint8)
                          if self.host_exchange else None)
        self.local_out = torch.empty(self.stage_bytes, dtype=torch.uint8, device=self.dev)
        self.send_offset = np.zeros(P, dtype=np.int64)
        self._stage = np.empty(self.stage_bytes, dtype=np.uint8)
        self._stage_off = [0]
        self._stage_used = 0
        self.done_flag = 0
        self.done_count = 0
        self._open = True
        return True

    def write(self, key: bytes, val: bytes = b""):
        """One pair.  key/val are the serialised bytes of the typed objects: a
        char* key is passed WITHOUT its NUL (added here, strlen+1 semantics of
        bytestream<char*>::to_bytes, src/serializer.h:145-157)."""
        rec = key + (b"\0" if self.layout.key_kind == capi.KEY_CSTR else b"")
        rec += val + (b"\0" if self.layout.val_kind == capi.VAL_CSTR else b"")
        self.write_records(np.frombuffer(rec, dtype=np.uint8), np.array([0, len(rec)], dtype=np.uint64))
        return True

    def write_records(self, blob, rec_off):
        """Many pairs already in wire form (emit order)."""
        n = len(rec_off) - 1
        i = 0
        while i < n:
            room = self.stage_bytes - self._stage_used
            base = int(rec_off[i])
            # as many whole records as fit the staging buffer
            j = int(np.searchsorted(rec_off, base + room, side="right")) - 1
            j = min(j, n, i + self.max_records - (len(self._stage_off) - 1))
            if j <= i:
                if self._stage_used == 0:
                    raise capi.MgsError(capi.ERR_RECORD_SIZE, "record larger than the staging buffer")
                self.flush()
                continue
            nb = int(rec_off[j]) - base
            self._stage[self._stage_used:self._stage_used + nb] = blob[base:base + nb]
            self._stage_off.extend((np.asarray(rec_off[i + 1:j + 1], dtype=np.int64) - base
                                    + self._stage_used).tolist())
            self._stage_used += nb
            i = j
        return True

    def make_progress(self, issue_new=False):
        self.flush()
        self.exchange_kv()

    def get_record_count(self):
        return self.kvcount

    def close(self):
        self.wait()
        self.packer.close()
        self._open = False

    # ---- the path ------------------------------------------------------------
    def flush(self):
        """Pack everything staged so far; every full partition ends a round."""
        n = len(self._stage_off) - 1
        if n == 0:
            return
        blob = torch.from_numpy(self._stage[:self._stage_used]).to(self.dev)
        d_off = None
        if not self.layout.fixed_len():
            d_off = torch.from_numpy(np.asarray(self._stage_off, dtype=np.uint32).view(np.int32)).to(self.dev)
        start = 0
        while start < n:
            r = self.packer.pack(blob, n, self.send_image, start=start, rec_off=d_off,
                                 init_offset=self.send_offset, local_out=self.local_out)
            self.send_offset[:] = r.send_offset
            self.kvcount += r.packed_records
            if r.local_bytes:  # out->write at emit time, src/collectiveshuffler.h:135-141
                loc = self.local_out[:r.local_bytes].cpu().numpy()
                self.out.write_records(loc, wire_offsets(self.layout, loc))
            if r.next < n:      # kv_to_bytes == -1: exchange, then retry (:148-160)
                if r.next == start and not r.local_records and int(self.send_offset.sum()) == 0:
                    raise capi.MgsError(capi.ERR_RECORD_SIZE, "KV size is larger than buf_size")
                self.exchange_kv()
            start = r.next
        self._stage_off = [0]
        self._stage_used = 0

    def exchange_kv(self):
        """src/collectiveshuffler.h:587-674."""
        P = self.shuffle_size
        self.send_bytes += int(self.send_offset.sum())
        if P == 1 or not dist.is_initialized():
            self.done_count = 1 if self.done_flag else 0
            self.send_offset[:] = 0
            self.shuffle_times += 1
            return
        if self.host_exchange:
            for d in range(P):
                so = int(self.send_offset[d])
                if so:
                    self.host_send[d * self.buf_size:d * self.buf_size + so].copy_(
                        self.send_image[d * self.buf_size:d * self.buf_size + so])
            src = self.host_send
        else:
            src = self.send_image
            if self.dev.type == "cuda":
                torch.cuda.current_stream(self.dev).synchronize()
        rc, self.done_count, got = exchange_round(src, self.send_offset.tolist(), self.buf_size,
                                                  self.recv_image, done=bool(self.done_flag),
                                                  group=self.group)
        self.send_offset[:] = 0
        if self.done_count >= P:
            return  # :623, shuffle_times is not advanced on the final vote
        self.recv_bytes += got
        if got:  # save_data, :563-585
            img = self.recv_image[:got].cpu().numpy()
            self.out.write_records(img, wire_offsets(self.layout, img))
        self.shuffle_times += 1

    def wait(self):
        """src/collectiveshuffler.h:528-561."""
        self.flush()
        while int(self.send_offset.sum()) > 0 or self.shuffle_times == 0:
            self.exchange_kv()
        self.done_flag = 1
        while True:
            self.exchange_kv()
            if self.done_count >= self.shuffle_size:
                break
