"""The exchange step of a shuffle round over torch.distributed.

Stands in for CollectiveShuffler::exchange_kv (src/collectiveshuffler.h:
587-674): an all-to-all of the per-destination byte counts, with -1 as the
in-band "I am done" vote (:599-621), then an all-to-all-v of the partition
bytes into a receive image that holds source 0's bytes, then source 1's, ...
without gaps (:630-645).

On GPUs the process group is "nccl" (= RCCL): the payload is one grouped batch
of point-to-point sends/receives straight out of the send image -- partition d
starts at d*buf_size and only its first send_offset[d] bytes travel -- so each
(src,dst) pair uses its own xGMI link and nothing is compacted first.  The
same code runs on CPU tensors over "gloo".
"""
import torch
import torch.distributed as dist


def exchange_counts(send_offset, done, group=None):
    """MPI_Alltoall of send_offset (src/collectiveshuffler.h:606-607).

    send_offset: int64 tensor [P] on the group's device (bytes per destination)
    done: this rank votes "finished" -> sends -1 to everybody (:599-603)
    Returns (recv_count int64 [P] with -1 mapped to 0, done_count)."""
    P = dist.get_world_size(group)
    out = torch.empty_like(send_offset)
    src = torch.full_like(send_offset, -1) if done else send_offset
    dist.all_to_all_single(out, src, group=group)
    assert out.numel() == P
    done_count = int((out == -1).sum().item())
    return torch.clamp(out, min=0), done_count


def exchange_payload(send_image, send_offset, recv_count, buf_size, recv_image,
                     group=None):
    """MPI_Alltoallv of the partitions (src/collectiveshuffler.h:630-645).

    send_image : uint8 [P*buf_size]; partition d at d*buf_size
    send_offset: host sequence of P ints (bytes to send to each rank)
    recv_count : host sequence of P ints (bytes coming from each rank)
    recv_image : uint8, at least sum(recv_count) bytes; filled src-ordered
    Returns the number of bytes received."""
    P = dist.get_world_size(group)
    me = dist.get_rank(group)
    ops, roff = [], 0
    for peer in range(P):
        rc = int(recv_count[peer])
        if rc > 0:
            if peer == me:  # cannot happen after the local bypass, kept for safety
                recv_image[roff:roff + rc].copy_(send_image[me * buf_size:me * buf_size + rc])
            else:
                ops.append(dist.P2POp(dist.irecv, recv_image[roff:roff + rc],
                                      dist.get_global_rank(group, peer) if group else peer,
                                      group=group))
        roff += rc
    for peer in range(P):
        so = int(send_offset[peer])
        if so > 0 and peer != me:
            ops.append(dist.P2POp(dist.isend,
                                  send_image[peer * buf_size:peer * buf_size + so],
                                  dist.get_global_rank(group, peer) if group else peer,
                                  group=group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        # MPI_Alltoallv returns when both images may be touched again.  On RCCL wait()
        # only orders the current stream behind the transfer; the packer writes the send
        # image from a stream of its own, so the host has to see the transfer end.
        if send_image.is_cuda:
            torch.cuda.current_stream(send_image.device).synchronize()
    return roff


def exchange_round(send_image, send_offset_host, buf_size, recv_image, done=False,
                   group=None):
    """One exchange_kv: counts, then payload.  Returns (recv_count list,
    done_count, bytes_received).  When every rank voted done nothing else
    moves (src/collectiveshuffler.h:623)."""
    P = dist.get_world_size(group)
    dev = send_image.device
    so = torch.as_tensor(list(send_offset_host), dtype=torch.int64, device=dev)
    rc, done_count = exchange_counts(so, done, group)
    rc_host = rc.cpu().tolist()
    if done_count >= P:
        return rc_host, done_count, 0
    sent = [0] * P if done else [int(x) for x in send_offset_host]
    got = exchange_payload(send_image, sent, rc_host, buf_size, recv_image, group)
    return rc_host, done_count, got


def exchange_rounds(send_images, image_stride, round_send_offset, buf_size, recv_images,
                    recv_stride, group=None, round_base=None):
    """The exchanges of a batch that was packed into many rounds at once
    (DevicePacker.pack_rounds): round k of every rank is the k-th exchange_kv of the
    reference (src/collectiveshuffler.h:587-674).  Ranks cut their rounds where their own
    data says, so they hold different numbers of rounds; a rank that has run out takes
    part in the remaining exchanges with empty partitions, as the reference's wait()
    does (:528-561).

    send_images      : uint8; image k at [k*image_stride :], partition d at d*buf_size
    round_send_offset: host int64 array [R_mine, P]: bytes per partition and round
    recv_images      : uint8; the receive image of exchange k is written at
                       [k*recv_stride :], source 0's bytes, then source 1's, ...
    round_base       : streams layout (DevicePacker.pack_streams): send_images holds P
                       destination streams of buf_size bytes, image_stride is ignored, and
                       round k's partition d begins at byte round_base[k, d] of stream d --
                       the all-to-all-v's send displacement is d*buf_size + round_base[k, d]
                       where the reference's is d*buf_size (src/collectiveshuffler.h:631-634)
    The counts of all rounds travel in one all-to-all (one row per round) instead of one
    per round; the payload goes round by round, each a grouped batch of point-to-point
    transfers straight out of the send image.  Returns (recv_count int64 [R, P], R)."""
    import numpy as np
    P = dist.get_world_size(group)
    me = dist.get_rank(group)
    dev = send_images.device
    so = np.asarray(round_send_offset, dtype=np.int64).reshape(-1, P)
    r_mine = so.shape[0]
    rmax = torch.tensor([r_mine], dtype=torch.int64, device=dev)
    dist.all_reduce(rmax, op=dist.ReduceOp.MAX, group=group)
    R = int(rmax.item())
    full = np.zeros((R, P), dtype=np.int64)
    full[:r_mine] = so
    # counts: rank d gets column d of every rank's matrix
    send = torch.from_numpy(np.ascontiguousarray(full.T)).to(dev)     # [P, R]
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    recv_count = recv.cpu().numpy().T.copy()                         # [R, P]: bytes from each source
    works = []
    for k in range(R):
        ops, roff = [], k * recv_stride
        for peer in range(P):
            rc = int(recv_count[k, peer])
            if rc > 0 and peer != me:
                ops.append(dist.P2POp(dist.irecv, recv_images[roff:roff + rc],
                                      dist.get_global_rank(group, peer) if group else peer, group=group))
            roff += rc
        for peer in range(P):
            sb = int(full[k, peer])
            if sb > 0 and peer != me:
                o = (k * image_stride + peer * buf_size if round_base is None
                     else peer * buf_size + int(round_base[k][peer]))
                ops.append(dist.P2POp(dist.isend, send_images[o:o + sb],
                                      dist.get_global_rank(group, peer) if group else peer, group=group))
        if ops:
            works.extend(dist.batch_isend_irecv(ops))
    for w in works:
        w.wait()
    if send_images.is_cuda:
        torch.cuda.current_stream(dev).synchronize()
    return recv_count, R


def exchange_streams(streams, stream_cap, round_send_offset, round_base, recv_streams, recv_cap, group=None):
    """All exchanges of a batch packed in the streams layout (DevicePacker.pack_streams) as
    ONE all-to-all-v.  In that layout the partitions a rank sends to destination d in its
    rounds 0, 1, 2, ... lie back to back in stream d, so everything d gets from this rank in
    the whole batch is one contiguous run: it travels as one message per (source,
    destination) pair -- P-1 sends and P-1 receives per rank instead of that many per round
    -- into one receive stream per source.  The reference's rounds survive as views: the
    counts of all rounds are exchanged as in exchange_rounds, and what source s sent in its
    round k is bytes [recv_base[k, s], + recv_count[k, s]) of receive stream s; walking k, then
    s, visits the received records in the order save_data hands them to the sink
    (src/collectiveshuffler.h:563-585, :637-639).

    streams          : uint8; stream d at [d*stream_cap :]
    round_send_offset: host int64 [R_mine, P]; round_base: host int64 [R_mine, P]
    recv_streams     : uint8; receive stream s at [s*recv_cap :]
    Returns (recv_count int64 [R, P], recv_base int64 [R, P], R)."""
    import numpy as np
    P = dist.get_world_size(group)
    me = dist.get_rank(group)
    dev = streams.device
    so = np.asarray(round_send_offset, dtype=np.int64).reshape(-1, P)
    base = np.asarray(round_base, dtype=np.int64).reshape(-1, P)
    r_mine = so.shape[0]
    rmax = torch.tensor([r_mine], dtype=torch.int64, device=dev)
    dist.all_reduce(rmax, op=dist.ReduceOp.MAX, group=group)
    R = int(rmax.item())
    full = np.zeros((R, P), dtype=np.int64)
    full[:r_mine] = so
    send = torch.from_numpy(np.ascontiguousarray(full.T)).to(dev)     # [P, R]
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    recv_count = recv.cpu().numpy().T.copy()                         # [R, P]
    recv_base = np.cumsum(recv_count, axis=0) - recv_count            # where round k of source s begins
    ops = []
    for peer in range(P):
        if peer == me:
            continue
        g = dist.get_global_rank(group, peer) if group else peer
        rc = int(recv_count[:, peer].sum())
        if rc > recv_cap:
            raise ValueError(f"receive stream of source {peer} needs {rc} bytes, has {recv_cap}")
        if rc > 0:
            ops.append(dist.P2POp(dist.irecv, recv_streams[peer * recv_cap:peer * recv_cap + rc], g, group=group))
    for peer in range(P):
        if peer == me:
            continue
        g = dist.get_global_rank(group, peer) if group else peer
        sb = int(full[:, peer].sum())
        if sb > 0:
            o = peer * stream_cap + int(base[0][peer])
            ops.append(dist.P2POp(dist.isend, streams[o:o + sb], g, group=group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if streams.is_cuda:
            torch.cuda.current_stream(dev).synchronize()
    return recv_count, recv_base, R


class StreamWindowExchange:
    """The all-to-all-v of a streams-layout batch in pieces, so that it overlaps with the
    packing that is still going on (the reference's overlap model is NBCollectiveShuffler,
    src/nbcollectiveshuffler.h:139-183, :222-387: pack into one buffer while others travel).

    The destination streams only ever grow at their ends, and a round that the packer has
    cut is final.  After every pack call the bytes of each stream that belong to finished
    rounds and have not travelled yet -- a window [lo, hi) per destination -- are posted
    here: their sizes go out in one small all-to-all (the counts exchange of
    src/collectiveshuffler.h:606-607, for a whole window), then one point-to-point message
    per (source, destination) pair, appended to the receiver's stream for that source.  The
    rounds stay views on both sides, exactly as in exchange_streams: the per-round counts are
    exchanged once at the end (finish), and what source s sent in its round k is bytes
    [recv_base[k, s], + recv_count[k, s]) of receive stream s.

    post() returns at once; the transfers of a window are waited for by the next post() or by
    finish() (one window in flight)."""

    def __init__(self, recv_streams, recv_cap, group=None, comm_stream=None):
        self.group = group
        self.P = dist.get_world_size(group)
        self.me = dist.get_rank(group)
        self.recv = recv_streams
        self.recv_cap = int(recv_cap)
        self.fill = [0] * self.P          # bytes received so far from each source
        self.works = []
        self.keep = []                    # host copies in flight (gloo)
        self.sent = [0] * self.P          # bytes posted so far to each destination
        # on GPUs the transfers are enqueued on a stream of their own, so that the pack
        # kernels enqueued meanwhile do not wait behind them
        self.comm_stream = comm_stream

    def _peer(self, r):
        return dist.get_global_rank(self.group, r) if self.group else r

    def _drain(self):
        for w in self.works:
            w.wait()
        self.works = []
        self.keep = []
        if self.comm_stream is not None:
            self.comm_stream.synchronize()

    def post(self, streams, stream_cap, lo, hi):
        """Send bytes [lo[d], hi[d]) of stream d (at streams[d * stream_cap:]) to rank d."""
        self._drain()
        P, me = self.P, self.me
        dev = self.recv.device
        sizes = [0 if d == me else max(0, int(hi[d]) - int(lo[d])) for d in range(P)]
        ctx = torch.cuda.stream(self.comm_stream) if self.comm_stream is not None else _NullCtx()
        with ctx:
            if self.comm_stream is not None:
                # what is sent was written by kernels of the caller's stream
                self.comm_stream.wait_stream(torch.cuda.current_stream(dev))
            send = torch.tensor(sizes, dtype=torch.int64, device=dev)
            got = torch.empty_like(send)
            dist.all_to_all_single(got, send, group=self.group)
            got = got.cpu().tolist()
            ops = []
            for s in range(P):
                rc = int(got[s])
                if s == me or rc == 0:
                    continue
                if self.fill[s] + rc > self.recv_cap:
                    raise ValueError(f"receive stream of source {s} needs {self.fill[s] + rc} bytes, has {self.recv_cap}")
                o = s * self.recv_cap + self.fill[s]
                ops.append(dist.P2POp(dist.irecv, self.recv[o:o + rc], self._peer(s), group=self.group))
                self.fill[s] += rc
            for d in range(P):
                if sizes[d] == 0:
                    continue
                o = d * int(stream_cap) + int(lo[d])
                piece = streams[o:o + sizes[d]]
                if piece.device != dev:       # gloo rehearsal: the streams live on the GPU
                    piece = piece.to(dev)
                    self.keep.append(piece)
                ops.append(dist.P2POp(dist.isend, piece, self._peer(d), group=self.group))
                self.sent[d] += sizes[d]
            if ops:
                self.works = dist.batch_isend_irecv(ops)

    def finish(self, round_send_offset):
        """Wait for the last window and exchange the per-round counts (as exchange_streams
        does).  Returns (recv_count int64 [R, P], recv_base int64 [R, P], R)."""
        import numpy as np
        self._drain()
        P = self.P
        dev = self.recv.device
        so = np.asarray(round_send_offset, dtype=np.int64).reshape(-1, P)
        rmax = torch.tensor([so.shape[0]], dtype=torch.int64, device=dev)
        dist.all_reduce(rmax, op=dist.ReduceOp.MAX, group=self.group)
        R = int(rmax.item())
        full = np.zeros((R, P), dtype=np.int64)
        full[:so.shape[0]] = so
        send = torch.from_numpy(np.ascontiguousarray(full.T)).to(dev)
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)
        recv_count = recv.cpu().numpy().T.copy()
        recv_base = np.cumsum(recv_count, axis=0) - recv_count
        for s in range(P):
            if s != self.me and int(recv_count[:, s].sum()) != self.fill[s]:
                raise RuntimeError(f"source {s}: {self.fill[s]} bytes arrived, its rounds add up to "
                                   f"{int(recv_count[:, s].sum())}")
        return recv_count, recv_base, R


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
