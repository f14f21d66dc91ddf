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
            image, offset = self._current_round()
            r = self.packer.pack(blob, n, image, start=start, rec_off=d_off,
                                 init_offset=offset, local_out=self.local_out)
            offset[:] = r.send_offset
            self.kvcount += r.packed_records
            if r.local_bytes:  # out->write at emit time, src/collectiveshuffler.h:135-141
                loc = self.local_out[:r.local_bytes].cpu().numpy()
                self._deliver(loc, self.shuffle_rank)
            if r.next < n:      # kv_to_bytes == -1: exchange, then retry (:148-160)
                if r.next == start and not r.local_records and int(offset.sum()) == 0:
                    raise capi.MgsError(capi.ERR_RECORD_SIZE, "KV size is larger than buf_size")
                self._round_full()
            self._after_pack()
            start = r.next
        self._stage_off = [0]
        self._stage_used = 0

    # hooks the non-blocking shuffler overrides
    def _current_round(self):
        return self.send_image, self.send_offset

    def _round_full(self):
        self.exchange_kv()

    def _after_pack(self):
        pass

    def _deliver(self, blob, src):
        """Hand wire-form records that came from rank `src` to the sink, in order."""
        offs = wire_offsets(self.layout, blob)
        if hasattr(self.out, "write_records_from"):
            self.out.write_records_from(src, blob, offs)
        else:
            self.out.write_records(blob, offs)

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


class StreamsGpuShuffler:
    """CollectiveShuffler for a map whose emitted records are device-resident batches (the
    output of mgs_tokenize_device, or any wire-form record stream in HBM): no record ever
    crosses the per-record write(key*, val*) boundary.

        open()
        write_device_records(d_blob, n, d_rec_off=None)   any number of times, emit order
        close()

    Every batch is packed with mgs_pack_streams_device into one stream per destination,
    going on where the batch before stopped, so the rounds are those CollectiveShuffler::write
    would cut over the whole sequence of records (src/collectiveshuffler.h:131-166).  close()
    moves everything in one all-to-all-v (exchange_streams) and hands the sink what the
    reference's `out` sees, in its order: the local records of round k (emitted during it,
    :135-141), then round k's receive image source by source (:563-585), for k = 0, 1, ...
    shuffle_times, kvcount and the byte counters are those of the reference's exchanges.
    The whole map's partitions stay in HBM until close(): size stream_bytes for them."""

    def __init__(self, out, layout, group=None, comm_buf_size=64 << 20, device=0, stream_bytes=1 << 30,
                 max_records=1 << 26, max_rounds=4096):
        if out is None:
            raise ValueError("Output shuffler cannot be NULL!")  # src/baseshuffler.h:35
        self.out, self.layout, self.group, self.device = out, layout, group, device
        self.shuffle_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.shuffle_rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.buf_size, self.unit_size, self.type_log_bytes = capi.geometry(comm_buf_size, self.shuffle_size)
        self.stream_cap = int(stream_bytes) // 16 * 16
        self.max_records, self.max_rounds = int(max_records), int(max_rounds)
        self.kvcount = self.shuffle_times = self.send_bytes = self.recv_bytes = 0

    def open(self):
        from .packer import DevicePacker
        P = self.shuffle_size
        self.dev = torch.device("cuda", self.device)
        self.packer = DevicePacker(self.layout, P, self.shuffle_rank, self.buf_size, self.max_records,
                                   device=self.device)
        self.streams = torch.empty(P * self.stream_cap, dtype=torch.uint8, device=self.dev)
        self.local = torch.empty(self.stream_cap, dtype=torch.uint8, device=self.dev)
        self._rounds, self._fill, self._base, self._lused = None, None, None, 0
        return True

    def write_device_records(self, d_blob, n, d_rec_off=None):
        from .packer import merge_stream_rounds
        start = 0
        while start < n:
            # (records too long for the one-sweep kernels: the call packs round by round with the
            # generic kernels, which take any record up to buf_size)
            r = self.packer.pack_streams(d_blob, n, self.streams, self.stream_cap, self.max_rounds, start=start,
                                         rec_off=d_rec_off, stream_fill=self._fill, open_round_base=self._base,
                                         local_out=self.local[self._lused:])
            if r.next == start and r.n_rounds >= self.max_rounds:
                raise capi.MgsError(capi.ERR_ARG, "max_rounds too small for this batch")
            if r.next < n and r.n_rounds < self.max_rounds and r.next - start < self.max_records:
                raise capi.MgsError(capi.ERR_ARG, "a destination stream is full: raise stream_bytes")
            self._lused += r.local_bytes
            self._base = r.round_base[-1].copy()
            self._fill = self._base + r.round_send_offset[-1]
            if self.shuffle_rank < len(self._fill):
                self._fill[self.shuffle_rank] = 0
            self._rounds = merge_stream_rounds(self._rounds, r)
            start = r.next
        return True

    def get_record_count(self):
        return self.kvcount

    def _deliver(self, blob, src):
        offs = wire_offsets(self.layout, blob)
        if hasattr(self.out, "write_records_from"):
            self.out.write_records_from(src, blob, offs)
        else:
            self.out.write_records(blob, offs)

    def close(self):
        from .exchange import exchange_streams
        P, me = self.shuffle_size, self.shuffle_rank
        R_mine = self._rounds
        so = R_mine.round_send_offset if R_mine is not None else np.zeros((1, P), dtype=np.int64)
        base = R_mine.round_base if R_mine is not None else np.zeros((1, P), dtype=np.int64)
        lbytes = R_mine.round_local_bytes if R_mine is not None else np.zeros(1, dtype=np.int64)
        self.kvcount = int(R_mine.packed_records) if R_mine is not None else 0
        self.send_bytes = int(so.sum())
        local = self.local[:self._lused].cpu().numpy()
        if P == 1 or not dist.is_initialized():
            self._deliver(local, me)
            self.shuffle_times = so.shape[0]
            self.packer.close()
            return
        backend = dist.get_backend(self.group)
        torch.cuda.current_stream(self.dev).synchronize()
        src = self.streams if backend == "nccl" else self.streams.cpu()
        # receive streams: a source sends this rank at most what its own stream for it holds
        need = torch.tensor([int(so.sum(axis=0).max())], dtype=torch.int64, device=src.device)
        dist.all_reduce(need, op=dist.ReduceOp.MAX, group=self.group)
        rcap = max(int(need.item()), 16)
        recv = torch.empty(P * rcap, dtype=torch.uint8, device=src.device)
        rc, rb, R = exchange_streams(src, self.stream_cap, so, base, recv, rcap, group=self.group)
        self.recv_bytes = int(rc.sum())
        self.shuffle_times = R
        host = recv.cpu().numpy()
        lpos = 0
        for k in range(R):
            if k < len(lbytes) and lbytes[k]:
                self._deliver(local[lpos:lpos + int(lbytes[k])], me)
                lpos += int(lbytes[k])
            for s in range(P):
                if s != me and rc[k, s]:
                    o = s * rcap + int(rb[k, s])
                    self._deliver(host[o:o + int(rc[k, s])], s)
        self.packer.close()


class CombineGpuShuffler(GpuShuffler):
    """CombineCollectiveShuffler (src/combinecollectiveshuffler.h) for a combiner
    declared as one of capi.COMBINE_* instead of a user callback: in a round only
    the first record of a key is written to its partition, later ones are combined
    into it (:91-151); a round ends before the first new key that does not fit
    (:122-127) and forgets its keys (:237); records that stay on this rank go to
    `out` untouched (:67-83).

    The device call works on a whole round at once, so the records of the current
    round stay staged until it ends.  When the staging buffer fills first, they
    are replaced by what the round has made of them so far -- the combined
    partitions, one after the other.  Inside a partition the order of first
    occurrence is kept, so partitions, round cuts and counters come out as in the
    reference."""

    def __init__(self, out, layout, op=capi.COMBINE_SUM_U64, **kw):
        super().__init__(out, layout, **kw)
        if layout.val_kind != capi.VAL_FIXED or layout.val_bytes != 8:
            raise ValueError("a declared combiner needs an 8-byte value")
        self.op = op
        # a refill may hold every partition, plus room for new records
        self.stage_bytes = max(self.stage_bytes, 2 * self.shuffle_size * self.buf_size)

    def open(self):
        super().open()
        self._loc_done = 0      # bytes of the round's local stream already handed to `out`
        self._round_keys = 0    # new keys of the round so far
        return True

    def write_records(self, blob, rec_off):
        n = len(rec_off) - 1
        i = 0
        while i < n:
            room = self.stage_bytes - self._stage_used
            base = int(rec_off[i])
            j = int(np.searchsorted(rec_off, base + room, side="right")) - 1
            j = min(j, n, i + self.max_records - (len(self._stage_off) - 1))
            if j <= i:
                if self._stage_used == 0:
                    raise capi.MgsError(capi.ERR_RECORD_SIZE, "record larger than the staging buffer")
                before = self._stage_used
                self._drain(final=False)
                if self._stage_used >= before:
                    raise capi.MgsError(capi.ERR_RECORD_SIZE, "staging buffer too small for a round")
                continue
            nb = int(rec_off[j]) - base
            self._stage[self._stage_used:self._stage_used + nb] = blob[base:base + nb]
            self._stage_off.extend((np.asarray(rec_off[i + 1:j + 1], dtype=np.int64) - base
                                    + self._stage_used).tolist())
            self._stage_used += nb
            i = j
        return True

    def flush(self):
        """Everything staged goes through the combiner; the round that is still open
        afterwards sits in the send image (and stays staged, see _drain)."""
        self._drain(final=True)

    def _drain(self, final):
        P, bs = self.shuffle_size, self.buf_size
        while True:
            n = len(self._stage_off) - 1
            if n == 0:
                return
            blob = torch.from_numpy(self._stage[:self._stage_used]).to(self.dev)
            d_off = None
            if not self.layout.fixed_len():
                d_off = torch.from_numpy(np.asarray(self._stage_off, dtype=np.uint32).view(np.int32)).to(self.dev)
            r = self.packer.combine(blob, n, self.send_image, start=0, rec_off=d_off,
                                    local_out=self.local_out, op=self.op)
            if r.local_bytes > self._loc_done:  # out->write at emit time, :67-83
                loc = self.local_out[self._loc_done:r.local_bytes].cpu().numpy()
                self._deliver(loc, self.shuffle_rank)
                self._loc_done = r.local_bytes
            self.send_offset[:] = r.send_offset
            self._round_keys = r.packed_records
            if r.next < n:  # a new key did not fit: the round ends here (:122-127)
                if r.next == 0:
                    raise capi.MgsError(capi.ERR_RECORD_SIZE, "KV size is larger than buf_size")
                self._end_round()
                cut = int(self._stage_off[r.next])
                rest = self._stage_used - cut
                self._stage[:rest] = self._stage[cut:self._stage_used].copy()
                self._stage_off = [o - cut for o in self._stage_off[r.next:]]
                self._stage_used = rest
                continue
            if final:
                return  # the open round is in the send image; its records stay staged
            # the round goes on but the staging buffer is full: keep what it has made so far
            img = self.send_image.cpu().numpy()
            used, offs = 0, [0]
            for d in range(P):
                so = int(r.send_offset[d])
                if so:
                    part = img[d * bs:d * bs + so]
                    self._stage[used:used + so] = part
                    offs.extend((wire_offsets(self.layout, part)[1:].astype(np.int64) + used).tolist())
                    used += so
            self._stage_off, self._stage_used, self._loc_done = offs, used, 0
            return

    def _end_round(self):
        self.kvcount += self._round_keys
        self._round_keys = 0
        self._loc_done = 0
        self.exchange_kv()

    def exchange_kv(self):
        # whatever is exchanged ends the round: its keys are forgotten (:237)
        if self._round_keys:
            self.kvcount += self._round_keys
            self._round_keys = 0
        super().exchange_kv()

    def make_progress(self, issue_new=False):
        self.flush()
        self._close_open_round()
        self.exchange_kv()

    def _close_open_round(self):
        self._stage_off, self._stage_used, self._loc_done = [0], 0, 0

    def wait(self):
        """src/collectiveshuffler.h:528-561 after garbage_collection() (:55)."""
        self.flush()
        self._close_open_round()
        super().wait()


class _MsgBuf:
    """ShuffleMsgBuf (src/nbcollectiveshuffler.h:22-34): one send image, one
    receive image and the state of the exchange that uses them."""
    START, PENDING, COMPLETE = 0, 1, 2

    def __init__(self, P, buf_size, dev, xdev, host_exchange):
        self.send_image = torch.empty(P * buf_size, dtype=torch.uint8, device=dev)
        self.recv_image = torch.empty(P * buf_size, dtype=torch.uint8, device=xdev)
        self.host_send = torch.empty(P * buf_size, dtype=torch.uint8) if host_exchange else None
        self.send_offset = np.zeros(P, dtype=np.int64)
        self.recv_count = None
        self.token = 0
        self.state = _MsgBuf.COMPLETE
        self.send_bytes = 0
        self.recv_bytes = 0
        self.a2a_work = None    # counts all-to-all in flight (MPI_Ialltoall, :237-239)
        self.a2a_in = None
        self.a2a_out = None
        self.a2av_works = None  # payload in flight (MPI_Ialltoallv, :347-351)


class NBGpuShuffler(GpuShuffler):
    """NBCollectiveShuffler (src/nbcollectiveshuffler.h): the same partition
    image per buffer, but 2..5 buffers, so that the exchange of a full round
    runs while the next round is being packed.  Counts travel on one process
    group and payloads on another (the reference's two duplicated
    communicators, :68-69), payloads strictly in token order (:318-359).

    The order in which a rank's sink sees its own records and the records of
    each peer is the same as with the blocking shuffler; how the streams of
    different sources interleave depends on when exchanges complete, as in the
    reference."""
    MIN_SBUF_COUNT = 2  # src/config.cpp:20-21
    MAX_SBUF_COUNT = 5

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        if dist.is_initialized() and self.shuffle_size > 1:
            ranks = ([dist.get_global_rank(self.group, r) for r in range(self.shuffle_size)]
                     if self.group is not None else None)
            self.a2a_group = dist.new_group(ranks)
            self.a2av_group = dist.new_group(ranks)
        else:
            self.a2a_group = self.a2av_group = None

    def open(self):
        super().open()
        self._xdev = self.recv_image.device
        self.msg = []
        self.cur_idx = 0
        self.a2a_token = 0
        self.a2av_token = 0
        self.pending_msg = 0
        self.round_log = []  # send_offset of every exchange started, in order (-1 = vote)
        self._p2p_blocking = dist.is_initialized() and dist.get_backend(self.group) != "nccl"
        for _ in range(self.MIN_SBUF_COUNT):
            self._insert_comm_buffer()
        return True

    def _insert_comm_buffer(self):  # :426-449
        self.msg.append(_MsgBuf(self.shuffle_size, self.buf_size, self.dev, self._xdev,
                                self.host_exchange))

    def _current_round(self):
        m = self.msg[self.cur_idx]
        return m.send_image, m.send_offset

    def _round_full(self):
        self.start_kv_exchange()

    def _after_pack(self):
        self.push_kv_exchange()  # after every record in the reference (:179)

    def make_progress(self, issue_new=False):  # :185-189
        self.flush()
        if issue_new and self.pending_msg == 0:
            self.start_kv_exchange()
        self.push_kv_exchange()

    def exchange_kv(self):
        raise RuntimeError("the non-blocking shuffler has no blocking exchange")

    # ---- :222-277 ----------------------------------------------------------------
    def start_kv_exchange(self):
        P = self.shuffle_size
        m = self.msg[self.cur_idx]
        if self.done_flag:
            m.send_offset[:] = -1
        else:
            m.send_bytes += int(m.send_offset.sum())
            self.send_bytes += int(m.send_offset.sum())
        self.shuffle_times += 1
        self.round_log.append(m.send_offset.copy())
        if self.a2a_group is not None:
            cdev = torch.device("cpu") if self._xdev.type == "cpu" else self.dev
            m.a2a_in = torch.as_tensor(m.send_offset, dtype=torch.int64).to(cdev)
            m.a2a_out = torch.empty_like(m.a2a_in)
            m.a2a_work = dist.all_to_all_single(m.a2a_out, m.a2a_in, group=self.a2a_group,
                                                async_op=True)
        else:  # a single rank talks to itself
            m.a2a_out = torch.as_tensor(m.send_offset.copy(), dtype=torch.int64)
            m.a2a_work = None
        m.recv_count = None
        m.state = _MsgBuf.START
        m.token = self.a2a_token
        self.a2a_token += 1
        self.pending_msg += 1
        self.push_kv_exchange()
        self.cur_idx = -1
        for i, b in enumerate(self.msg):
            if b.state == _MsgBuf.COMPLETE:
                self.cur_idx = i
                break
        if self.cur_idx == -1:
            if len(self.msg) < self.MAX_SBUF_COUNT:
                self._insert_comm_buffer()
                self.cur_idx = len(self.msg) - 1
            else:
                self.cur_idx = 0
                while True:
                    self.push_kv_exchange()
                    if self.msg[self.cur_idx].state == _MsgBuf.COMPLETE:
                        break
                    self.cur_idx = (self.cur_idx + 1) % len(self.msg)

    # ---- :279-387 ----------------------------------------------------------------
    def push_kv_exchange(self):
        P = self.shuffle_size
        for m in self.msg:
            if m.state == _MsgBuf.COMPLETE:
                continue
            if m.state == _MsgBuf.START:
                if m.recv_count is None and (m.a2a_work is None or m.a2a_work.is_completed()):
                    if m.a2a_work is not None:
                        m.a2a_work.wait()
                    rc = m.a2a_out.cpu().numpy().astype(np.int64)
                    self.done_count = int((rc == -1).sum())
                    m.recv_bytes += int(rc[rc != -1].sum())
                    m.recv_count = rc
                    m.a2a_work = None
                if self.done_count == P and m.recv_count is not None:
                    m.state = _MsgBuf.COMPLETE
                    self.pending_msg -= 1
                    self._reset(m)
                    continue
                if (m.token == self.a2av_token and m.recv_count is not None
                        and self.done_count < P):
                    m.send_offset[m.send_offset == -1] = 0
                    m.recv_count[m.recv_count == -1] = 0
                    self._start_payload(m)
                    self.recv_bytes += int(m.recv_count.sum())
                    m.state = _MsgBuf.PENDING
                    self.a2av_token += 1
            if m.state == _MsgBuf.PENDING and not any(
                    o.state == _MsgBuf.PENDING and o.token < m.token for o in self.msg):
                # Rounds reach `out` in the order they were sent (the reference hands completed
                # exchanges over in buffer-index order, :362-383, which can swap two rounds).
                # gloo's point-to-point requests only complete inside wait(): block there.
                # Payloads are posted in token order on every rank, so this cannot deadlock;
                # it only gives up the overlap, which matters on RCCL, where polling works.
                if self._p2p_blocking or all(w.is_completed() for w in m.a2av_works):
                    for w in m.a2av_works:
                        w.wait()
                    m.a2av_works = None
                    self._save_data(m)
                    self._reset(m)
                    m.state = _MsgBuf.COMPLETE
                    self.pending_msg -= 1

    def _reset(self, m):
        m.token = 0
        m.send_bytes = 0
        m.recv_bytes = 0
        m.send_offset[:] = 0
        m.recv_count = None

    def _start_payload(self, m):
        P, me, bs = self.shuffle_size, self.shuffle_rank, self.buf_size
        if self.a2av_group is None:
            m.a2av_works = []
            return
        if self.host_exchange:
            for d in range(P):
                so = int(m.send_offset[d])
                if so:
                    m.host_send[d * bs:d * bs + so].copy_(m.send_image[d * bs:d * bs + so])
            src = m.host_send
        else:
            src = m.send_image
            if self.dev.type == "cuda":
                torch.cuda.current_stream(self.dev).synchronize()
        ops, roff = [], 0
        for peer in range(P):
            rc = int(m.recv_count[peer])
            if rc > 0 and peer != me:
                ops.append(dist.P2POp(dist.irecv, m.recv_image[roff:roff + rc],
                                      dist.get_global_rank(self.a2av_group, peer),
                                      group=self.a2av_group))
            roff += rc
        for peer in range(P):
            so = int(m.send_offset[peer])
            if so > 0 and peer != me:
                ops.append(dist.P2POp(dist.isend, src[peer * bs:peer * bs + so],
                                      dist.get_global_rank(self.a2av_group, peer),
                                      group=self.a2av_group))
        m.a2av_works = dist.batch_isend_irecv(ops) if ops else []

    def _save_data(self, m):  # :400-424
        roff = 0
        for src in range(self.shuffle_size):
            rc = int(m.recv_count[src])
            if rc:
                self._deliver(m.recv_image[roff:roff + rc].cpu().numpy(), src)
            roff += rc

    def wait_all(self):  # :396-399
        while self.pending_msg > 0:
            self.push_kv_exchange()

    def wait(self):  # :192-220
        self.flush()
        m = self.msg[self.cur_idx]
        m.send_bytes += int(m.send_offset.sum())
        if m.send_bytes > 0:
            self.start_kv_exchange()
        self.wait_all()
        self.done_flag = 1
        while True:
            self.start_kv_exchange()
            self.wait_all()
            if self.done_count >= self.shuffle_size:
                break
