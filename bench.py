#!/usr/bin/env python3
"""Headline benchmark: device-resident hash-partition + pack of <u64,u64>
records into 8 destination partitions (the 1e9-record fixed-width configuration).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--records R]

A step is one pass of the hot path (mgs_pack_device: hash -> destination ->
stable scatter-pack) over one batch of R synthetic records already resident in
HBM.  N = 1: R = 1e9 records, P = 8 partitions, no local bypass.  N > 1 (one
process per GPU under torch.distributed.run): every rank packs its own R
records for P = N destinations with the local bypass on and then exchanges the
partitions with RCCL point-to-point all-to-all; the step covers both.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel against
HBM: algorithmic bytes = 32 B/record (read 16 + write 16) over the kernel's
device time from HIP events recorded on its own stream.  `cpu_baseline` is the
CPU oracle (a plain-C port of the reference algorithm) timed on this box: 16
processes, one per core, each packing its slice of the same record stream; the
one-core figure is kept beside it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E peak
XGMI_LINK_GBS = 153.0       # one directed xGMI link
ALGO_BYTES_PER_RECORD = 32  # read the 16-byte record once + write it once


def cpu_baseline(records_dev, n_total, target_seconds=12.0):
    """Time the oracle's CollectiveShuffler::write loop on one host core, on a
    prefix of the very records the GPU packed."""
    import ctypes as C
    import numpy as np
    from oracle import oracle as O
    orc = O.Oracle()
    lay = O.Layout.make(O.KEY_FIXED, 8, O.VAL_FIXED, 8)
    P = 8

    def run(n):
        blob = records_dev[:16 * n].cpu().numpy()
        buf = 16 * (n // P + n // 64 + 4096)
        img = np.zeros(P * buf, dtype=np.uint8)  # pages touched, like the reference's open()
        so = np.zeros(P, dtype=np.int64)
        lb, lr, pr = C.c_int64(), C.c_uint64(), C.c_uint64()
        t0 = time.perf_counter()
        nxt = orc.L.mo_pack_round(C.byref(lay), blob.ctypes.data, None, 16, n, 0, P, -1, buf,
                                  img.ctypes.data, so.ctypes.data, None, 0,
                                  C.byref(lb), C.byref(lr), C.byref(pr))
        dt = time.perf_counter() - t0
        assert nxt == n
        return dt

    n0 = min(1 << 22, n_total)
    dt0 = run(n0)
    n = int(min(max(n0, n0 * target_seconds / max(dt0, 1e-6)), n_total))
    dt = run(n)
    return {"value": n / dt, "unit": "records/s", "cores": 1, "kind": "port",
            "GBps": n * ALGO_BYTES_PER_RECORD / dt / 1e9,
            "sample": f"first {n} of the benchmark's own <u64,u64> records, P=8, one round, "
                      f"oracle/mimir_oracle.c mo_pack_round (C restatement of "
                      f"CollectiveShuffler::write) on one host core, {dt:.1f} s"}


def cpu_baseline_ref_pieces(records_dev, n_total, target_seconds=6.0):
    """The same loop with the reference's own compiled pieces -- Serializer::get_hash_code ->
    hashlittle, modulo, Serializer::kv_to_bytes (oracle/_ref/libmimir_ref_ser.so, built in the
    build container from the sources under /root/reference) -- on one host core, on a prefix
    of the records the GPU packed, checked against the port's partitions.  None where the
    library is absent or does not load."""
    import numpy as np
    from oracle import oracle as O
    try:
        ref = O.ReferenceSerializer()
    except (OSError, FileNotFoundError) as e:
        print(f"# reference pieces not timed: {e}", file=sys.stderr)
        return None
    P = 8

    def run(n, check=False):
        blob = records_dev[:16 * n].cpu().numpy()
        buf = 16 * (n // P + n // 64 + 4096)
        img = np.zeros(P * buf, dtype=np.uint8)
        so = np.zeros(P, dtype=np.int64)
        t0 = time.perf_counter()
        done = ref.write_loop_u64_u64(blob, n, P, img, buf, so)
        dt = time.perf_counter() - t0
        assert done == n, (done, n, so.tolist(), buf)
        if check:  # the port gives the same partitions
            orc = O.Oracle()
            lay = O.Layout.make(O.KEY_FIXED, 8, O.VAL_FIXED, 8)
            want = orc.pack_round(lay, blob, None, n, 0, P, -1, buf)
            assert want["send_offset"].tolist() == so.tolist()
            for d in range(P):
                assert np.array_equal(img[d * buf:d * buf + int(so[d])], want["partitions"][d])
        return dt

    n0 = min(1 << 21, n_total)
    dt0 = run(n0, check=True)
    # (kv_to_bytes takes the room left in a partition as an int: partitions stay below 2 GiB)
    n = int(min(max(n0, n0 * target_seconds / max(dt0, 1e-6)), n_total, 400_000_000))
    dt = run(n)
    return {"value": n / dt, "unit": "records/s", "cores": 1, "kind": "ref-pieces",
            "GBps": n * ALGO_BYTES_PER_RECORD / dt / 1e9,
            "sample": f"first {n} of the benchmark's own <u64,u64> records, P=8, one round: the reference's "
                      f"compiled Serializer::get_hash_code (hashlittle) % P and Serializer::kv_to_bytes per "
                      f"record (oracle/ref_probe_ser.cpp ser_write_loop_u64_u64) on one host core, {dt:.1f} s; "
                      f"partitions equal the port's on the first {n0}"}


def cpu_baseline_multicore(n_total):
    """The same loop on every physical core this job may use (one process per
    core, each a rank with its own slice and its own send image), run as a child
    process that never touches the GPU.  None if it cannot be run."""
    import subprocess
    cpus = host_cpus()
    procs = max(1, min(cpus["physical"], cpus["share"]))
    cmd = [sys.executable, os.path.join(ROOT, "tools", "cpu_pack_baseline.py"),
           "--procs", str(procs), "--records", str(int(min(n_total, 1_000_000_000)))]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, check=True)
        out = json.loads(r.stdout.strip().splitlines()[-1])
        out["host_cpus"] = cpus
        return out
    except Exception as e:  # the single-core figure stands in
        print(f"# multi-core CPU baseline not run: {e}", file=sys.stderr)
        return None


def committed_traffic(kernel_substr, n, rounds=1):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3
    PMC passes (FETCH_SIZE doubled: gfx950 tallies 128-byte reads as 64; both
    counters are in KiB), if such a profile exists for this kernel and size."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
    except OSError:
        return None
    if t.get("records") != n or kernel_substr not in t.get("kernel", "") or t.get("rounds", 1) != rounds:
        return None
    return t.get("hbm_bytes_per_launch")


def launch_ranks(n_ranks):
    """`--gpus N` with N > 1 and no launcher around us: start the N ranks ourselves,
    one process per GPU under torch.distributed.run, before this process has made
    any GPU call (it never does), and relay rank 0's JSON line."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for line in proc.stdout:
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    return proc.wait()


def host_cpus():
    """What the host offers and what this job may use: logical CPUs, physical
    cores (distinct (package, core) pairs of /proc/cpuinfo), and the job's share
    (scheduler affinity, cgroup CPU quota, OMP_NUM_THREADS when a harness sets it)."""
    logical = os.cpu_count() or 1
    cores = set()
    try:
        pkg = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    pkg = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if core is not None:
                        cores.add((pkg, core))
                    pkg = core = None
    except OSError:
        pass
    physical = len(cores) or logical
    share = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else logical
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            share = min(share, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    try:
        share = min(share, max(1, int(os.environ["OMP_NUM_THREADS"])))
    except (KeyError, ValueError):
        pass
    return {"logical": logical, "physical": physical, "share": share}


def exchange_sums(view16):
    """(records, wrap-around sum of the key words, of the value words) of a run of <u64,u64> records."""
    import torch
    if view16.numel() == 0:
        return [0, 0, 0]
    w = view16.view(torch.int64).view(-1, 2)
    return [int(w.shape[0]), int(w[:, 0].sum().item()), int(w[:, 1].sum().item())]


def verify_exchange(dist, torch, np, capi, DevicePacker, lay, P, me, streams, stream_cap, so_tot, recv, recv_cap,
                    recv_tot, dev, backend):
    """The exchange of one step, end to end (16-byte <u64,u64> records): what every source says
    it sent to this rank -- record count and wrap-around sums of keys and values -- is what
    arrived from it; every record that arrived belongs here by the library's own hash (the
    received stream packed again for P destinations lands in partition `me` alone); and a
    source's records arrive in its emit order (the value is the emit index).  Returns a dict
    for the bench line; raises on a mismatch."""
    mine = torch.zeros((P, 3), dtype=torch.int64)
    for d in range(P):
        if d != me:
            mine[d] = torch.tensor(exchange_sums(streams[d * stream_cap:d * stream_cap + int(so_tot[d])]))
    everyone = [torch.zeros((P, 3), dtype=torch.int64, device=dev if backend == "nccl" else "cpu") for _ in range(P)]
    dist.all_gather(everyone, mine.to(everyone[0].device))
    checked = 0
    for s in range(P):
        if s == me:
            continue
        nbytes = int(recv_tot[s])
        got = recv[s * recv_cap:s * recv_cap + nbytes]
        want = everyone[s][me].cpu().tolist()
        have = exchange_sums(got)
        if have != want:
            raise RuntimeError(f"rank {me}: from source {s} arrived (records, key sum, value sum) {have}, it sent {want}")
        if nbytes:
            vals = got.view(torch.int64).view(-1, 2)[:, 1]
            if not bool((vals[1:] > vals[:-1]).all().item()):
                raise RuntimeError(f"rank {me}: records of source {s} are not in its emit order")
            chk = DevicePacker(lay, P, -1, nbytes, max_records=nbytes // 16, device=dev)
            img = torch.empty(P * nbytes, dtype=torch.uint8, device=f"cuda:{dev}")
            res = chk.pack(got, nbytes // 16, img)
            chk.close()
            del img
            if int(res.send_offset[me]) != nbytes or int(res.send_offset.sum()) != nbytes:
                raise RuntimeError(f"rank {me}: records from source {s} that hash to another rank arrived")
        checked += have[0]
    return {"verified": True, "records_checked": checked,
            "checks": "per (source, destination) pair: record count, wrap-around key and value sums as sent = as "
                      "received; destination of every received record = this rank; values of a source increasing"}


def parse_size(text):
    """Sizes as MIMIR_COMM_SIZE takes them (src/mimir.cpp:75-131): a number with an
    optional b/k/m/g suffix."""
    t = str(text).strip().lower().rstrip("b")
    mult = {"k": 1 << 10, "m": 1 << 20, "g": 1 << 30}.get(t[-1:], 1)
    return int(float(t[:-1] if t[-1:] in "kmg" else t) * mult)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--records", type=float, default=1e9)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--algo", default="auto", choices=["auto", "generic", "onesweep16", "onesweepv"])
    ap.add_argument("--workload", default="u64", choices=["u64", "words", "join", "octree"],
                    help="u64: the headline configuration; words / join / octree: the other record "
                         "shapes (host-generated, default 5e7 records per GPU)")
    ap.add_argument("--level", type=int, default=12, help="octree: key characters per record")
    ap.add_argument("--comm-size", default=None,
                    help="MIMIR_COMM_SIZE: the send buffer of the reference (64M by default there); the "
                         "partitions get the reference's geometry and the batch is cut into rounds as "
                         "CollectiveShuffler::write cuts it.  0: one round sized to hold the whole batch. "
                         "Default: 64M.")
    ap.add_argument("--layout", default="streams", choices=["streams", "images"],
                    help="how the rounds of a batch lie in memory.  streams: one stream per destination, the "
                         "rounds are views into it (mgs_pack_streams_device: the sweep never waits for a round "
                         "cut); images: one send image per round (mgs_pack_rounds_device).  Same rounds, same "
                         "bytes per partition either way.")
    ap.add_argument("--chunks", type=int, default=1,
                    help="streams layout: a step packs the device-resident batch this many times over, every call "
                         "going on in the same destination streams where the one before stopped (stream_fill / "
                         "open_round_base) -- a step of chunks x records records.  Variable-width batches stay "
                         "under 4 GiB (32-bit record offsets); this is how 1e8..1e9 of them are packed.")
    ap.add_argument("--split", type=int, default=0,
                    help="N > 1, streams layout: the batch is packed in this many calls of records / split records "
                         "each, every call going on in the same streams (the result is that of one call), and what "
                         "a call has finished -- the rounds it cut -- travels while the next call packs "
                         "(StreamWindowExchange on a stream of its own).  Default: 4 for N > 1; 1 = pack everything, "
                         "then exchange.")
    ap.add_argument("--verify", action="store_true",
                    help="N > 1: one more, untimed step whose exchange is checked end to end: record counts and "
                         "wrap-around sums of keys and values per (source, destination) pair as sent and as "
                         "received, every received record's destination (the library's own hash) = this rank, "
                         "values of a source in emit order.  On by default for the u64 workload.")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--exchange", default="batch", choices=["batch", "rounds"],
                    help="N > 1, streams layout: batch = everything a destination gets from this rank in the "
                         "whole batch as ONE message (its rounds lie back to back in the stream; the rounds come "
                         "out as views on the receive side too); rounds = one grouped send/recv batch per "
                         "round of the reference.  The image layout always exchanges round by round.")
    args = ap.parse_args()
    if args.workload != "u64" and args.records == 1e9:
        args.records = 5e7
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist
    from mimir_amd import capi
    from mimir_amd.packer import DevicePacker, synth_u64_records

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # MGS_BENCH_FORCE_DIST=1: take the N > 1 branch with a world of one rank too (on a one-GPU box this is
    # the only way to execute the RCCL process group, its collectives and the exchange code at all)
    distributed = world > 1 or os.environ.get("MGS_BENCH_FORCE_DIST") == "1"
    # MGS_BENCH_BACKEND=gloo is the rehearsal mode for boxes with fewer GPUs than
    # ranks: ranks share the devices round-robin and the partitions travel through
    # host memory.  Its numbers say nothing about xGMI and are labelled as such.
    backend = os.environ.get("MGS_BENCH_BACKEND", "nccl")
    dev = local_rank % max(1, torch.cuda.device_count()) if backend == "gloo" else local_rank
    torch.cuda.set_device(dev)
    if distributed and backend == "gloo":
        dist.init_process_group("gloo")
    elif distributed:
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))

    n = int(args.records)
    P = 8 if not distributed else world
    self_rank = -1 if not distributed else rank
    comm_size = parse_size(args.comm_size if args.comm_size is not None else "64M")
    algo = {"auto": capi.ALGO_AUTO, "generic": capi.ALGO_GENERIC,
            "onesweep16": capi.ALGO_ONESWEEP16, "onesweepv": capi.ALGO_ONESWEEPV}[args.algo]
    rec_off = None
    if args.workload == "u64":
        lay = capi.Layout.make(capi.KEY_FIXED, 8, capi.VAL_FIXED, 8)
        records = synth_u64_records(n, seed=42 + rank, first=rank * n, device=dev)
        total_bytes, algo_bytes = 16 * n, n * ALGO_BYTES_PER_RECORD
        hot_share = 1.0 / P
        what, data = "configs[1]: fixed-width <uint64,uint64> records", \
            "synthetic (seeded splitmix64 keys, value = record index)"
    else:
        from mimir_amd import datagen
        if args.workload == "words":   # gen_words-like corpus: Zipf over a vocabulary, word\0 + u64 count
            blob, offs, lt = datagen.word_records(n, 1_000_000, 1.0, 6, 1.0, seed=44 + rank)
            what = "configs[3]: wordcount records (NUL-terminated word + uint64), Zipf(1.0) over 1e6 words"
        elif args.workload == "join":  # gen_tables-like: word\0 + {tag, pad, int64}
            blob, offs, lt = datagen.join_records(n, seed=45 + rank)
            what = "configs[4]: join records (NUL-terminated key + 16-byte tagged value), Zipf keys"
        else:                          # octree level pass: `level` key characters + u64 count
            blob, offs, lt = datagen.octree_records(n, level=args.level, seed=43 + rank)
            what = f"configs[2]: octree level-{args.level} records ({args.level}-char octant key + uint64)"
        lay = capi.Layout.make(*lt)
        total_bytes = int(blob.shape[0])
        records = torch.from_numpy(blob).to(f"cuda:{dev}")
        if offs is not None:
            rec_off = torch.from_numpy(offs.astype(np.uint32).view(np.int32)).to(f"cuda:{dev}")
        hot_share = min(1.0, 4.0 / P)  # skewed keys: room for four times the even share
        algo_bytes = 2 * total_bytes + (4 * (n + 1) if offs is not None else 0)
        data = "synthetic (seeded host generator, mimir_amd/datagen.py)"

    if comm_size:
        # the reference's geometry (src/collectiveshuffler.h:33-53); rounds are cut by the data
        buf_size, _, tlb = capi.geometry(comm_size, P)
        if tlb != 0:
            raise SystemExit("--comm-size above 1 GiB needs MPI count units of more than one byte "
                             "(type_log_bytes > 0), which this path does not take")
        max_rounds = int(total_bytes * hot_share / buf_size * 1.15) + 8
        rounds_note = f"MIMIR_COMM_SIZE={comm_size} -> {buf_size}-byte partitions, rounds cut by the data"
    else:
        # one round holds the whole batch: capacity = expected share + slack
        buf_size = (int(total_bytes * hot_share) + total_bytes // 64 + 65536) // 16 * 16
        max_rounds = 1
        rounds_note = "one round holds the batch"
    stride = P * buf_size
    streams_layout = bool(comm_size) and args.layout == "streams"
    chunks = max(1, args.chunks) if streams_layout else 1
    max_rounds = max_rounds * chunks
    pk = DevicePacker(lay, P, self_rank, buf_size, max_records=n, device=dev)
    if streams_layout:
        # one stream per destination, sized for its share of the batch plus slack
        stream_cap = (chunks * int(total_bytes * hot_share) + total_bytes // 64 + 65536) // 16 * 16
        images = torch.empty(P * stream_cap, dtype=torch.uint8, device=f"cuda:{dev}")
        rounds_note += "; rounds are views into one stream per destination"
    else:
        images = torch.empty(max_rounds * stride, dtype=torch.uint8, device=f"cuda:{dev}")
    local = recv = None
    if distributed:
        local = torch.empty(chunks * int(total_bytes * hot_share) + total_bytes // 64 + 65536, dtype=torch.uint8,
                            device=f"cuda:{dev}")
        batch_exchange = streams_layout and args.exchange == "batch"
        # batch: one receive stream per source; else one receive image per round
        recv = torch.empty(P * stream_cap if batch_exchange else max_rounds * stride, dtype=torch.uint8,
                           device=f"cuda:{dev}")
        from mimir_amd.exchange import StreamWindowExchange, exchange_rounds, exchange_streams
    split = 1
    if distributed and streams_layout and args.exchange == "batch" and chunks == 1:
        split = args.split if args.split > 0 else 4
        split = max(1, min(split, n))
    comm_stream = torch.cuda.Stream(device=dev) if (distributed and backend == "nccl") else None

    kernel_ms, pack_s, stats = [], [], {"sent_max": 0, "rounds": 0, "exchanges": 0}

    def step():
        t0 = time.perf_counter()
        if streams_layout:
            from mimir_amd.packer import merge_stream_rounds
            r, fill, base, lused, kms = None, None, None, 0, 0.0
            calls = [(0, n)] * chunks if split == 1 else [(i * n // split, (i + 1) * n // split) for i in range(split)]
            windows = None
            if split > 1:
                # what a call has finished travels while the next call packs
                dst = torch.empty(recv.shape, dtype=torch.uint8) if backend == "gloo" else recv
                windows = StreamWindowExchange(dst, stream_cap, comm_stream=comm_stream)
                sent = np.zeros(P, dtype=np.int64)
            t_pack = 0.0
            for ci, (a, b) in enumerate(calls):
                tp = time.perf_counter()
                rc_ = pk.pack_streams(records, b, images, stream_cap, max_rounds, start=a, rec_off=rec_off,
                                      stream_fill=fill, open_round_base=base,
                                      local_out=None if local is None else local[lused:])
                t_pack += time.perf_counter() - tp
                if rc_.next != b:
                    raise RuntimeError(f"{rc_.n_rounds} rounds used up at record {rc_.next} of {b}: raise max_rounds")
                kms += pk.last_kernel_ms()
                lused += rc_.local_bytes
                base = rc_.round_base[-1].copy()
                fill = base + rc_.round_send_offset[-1]
                if self_rank >= 0:
                    fill[self_rank] = 0
                r = merge_stream_rounds(r, rc_)
                if windows is not None:
                    # the rounds this call cut are final; its last round is still open unless it was the last call
                    hi = fill if ci == len(calls) - 1 else base
                    hi = np.maximum(hi, sent)
                    windows.post(images, stream_cap, sent, hi)
                    sent = hi.copy()
            pack_s.append(t_pack if windows is not None else time.perf_counter() - t0)
            kernel_ms.append(kms)
            stats["rounds"] = r.n_rounds
            if windows is not None:
                rcw, rbw, R = windows.finish(r.round_send_offset)
                if backend == "gloo":
                    recv.copy_(dst)
                stats["exchanges"] = R
                stats["sent_max"] = max(stats["sent_max"], int(r.round_send_offset.sum(axis=0).max()))
                stats["last"] = (r, rcw, rbw)
                return r
        else:
            r = pk.pack_rounds(records, n, images, stride, max_rounds, rec_off=rec_off, local_out=local, algo=algo)
            pack_s.append(time.perf_counter() - t0)
            if r.next != n:
                raise RuntimeError(f"{r.n_rounds} rounds used up at record {r.next} of {n}: raise max_rounds")
            kernel_ms.append(pk.last_kernel_ms())
            stats["rounds"] = r.n_rounds
        if distributed:
            so = r.round_send_offset
            part, bases = (stream_cap, r.round_base) if streams_layout else (buf_size, None)
            src_img = images.cpu() if backend == "gloo" else images   # gloo rehearsal: through host memory
            dst_img = torch.empty(recv.shape, dtype=torch.uint8) if backend == "gloo" else recv
            if batch_exchange:
                rc, rb, R = exchange_streams(src_img, stream_cap, so, bases, dst_img, stream_cap)
                stats["last"] = (r, rc, rb)
            else:
                rc, R = exchange_rounds(src_img, stride, so, part, dst_img, stride, round_base=bases)
            if backend == "gloo":
                recv.copy_(dst_img)
            stats["exchanges"] = R
            stats["sent_max"] = max(stats["sent_max"], int(so.sum(axis=0).max()))
        return r

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    kernel_ms.clear()
    pack_s.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64,
                         device="cpu" if backend == "gloo" else f"cuda:{dev}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- N > 1, after the timed region: the two exchange protocols side by side, what the
    # overlap hid, and the end-to-end check of what arrived (run below, once the line that holds
    # the measurement is complete, under a watchdog: a hang or an error in here must not
    # lose the measurement)
    extra = {}

    def post_timing_extras():
        r_last = stats["last"][0]
        so_l, base_l = r_last.round_send_offset, r_last.round_base
        src_img = images.cpu() if backend == "gloo" else images

        def timed(fn, reps=2):
            best = None
            for _ in range(reps):
                barrier()
                t1 = time.perf_counter()
                fn()
                barrier()
                d1 = time.perf_counter() - t1
                best = d1 if best is None else min(best, d1)
            tt = torch.tensor([best], dtype=torch.float64, device="cpu" if backend == "gloo" else f"cuda:{dev}")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item()) * 1e3

        dst_b = torch.empty(recv.shape, dtype=torch.uint8) if backend == "gloo" else recv
        extra["batch_ms"] = timed(lambda: exchange_streams(src_img, stream_cap, so_l, base_l, dst_b, stream_cap))
        rounds_bytes = (int(so_l.shape[0]) + 8) * stride
        free_b = torch.cuda.mem_get_info(dev)[0]
        if backend == "gloo" or rounds_bytes < free_b // 2:
            dst_r = torch.empty(rounds_bytes, dtype=torch.uint8, device="cpu" if backend == "gloo" else f"cuda:{dev}")
            rmax = torch.tensor([so_l.shape[0]], dtype=torch.int64, device=dst_r.device)
            dist.all_reduce(rmax, op=dist.ReduceOp.MAX)
            if int(rmax.item()) + 1 <= rounds_bytes // stride:
                extra["rounds_ms"] = timed(lambda: exchange_rounds(src_img, stride, so_l, stream_cap, dst_r, stride,
                                                                   round_base=base_l))
            del dst_r
        verify = (args.verify or args.workload == "u64") and not args.no_verify and args.workload == "u64"
        if verify:
            # one more exchange of the last step's streams into a cleared receive buffer
            recv.zero_()
            dst_v = torch.empty(recv.shape, dtype=torch.uint8) if backend == "gloo" else recv
            rc_v, _, _ = exchange_streams(src_img, stream_cap, so_l, base_l, dst_v, stream_cap)
            if backend == "gloo":
                recv.copy_(dst_v)
            extra.update(verify_exchange(dist, torch, np, capi, DevicePacker, lay, P, rank, images, stream_cap,
                                         so_l.sum(axis=0), recv, stream_cap, rc_v.sum(axis=0), dev, backend))
            # ... and the windows of the timed steps delivered the same bytes: the last timed
            # step's receive streams were overwritten above, so compare a fresh windowed step
            if split > 1:
                keep = recv.clone()
                step()
                torch.cuda.synchronize()
                for s_ in range(P):
                    nb_ = int(rc_v[:, s_].sum())
                    if s_ != rank and not torch.equal(keep[s_ * stream_cap:s_ * stream_cap + nb_],
                                                      recv[s_ * stream_cap:s_ * stream_cap + nb_]):
                        raise RuntimeError(f"rank {rank}: the windowed exchange delivered other bytes from source {s_}")
                extra["windows_equal_batch"] = True
                del keep


    ms_per_step = dt / args.steps * 1e3
    value = n * chunks * world * args.steps / dt
    algo_bytes *= chunks
    in_kernel = (pk.last_algo() in ("onesweep16", "onesweepd") or max_rounds == 1 or stats["rounds"] == 1
                 or streams_layout)
    # one kernel covers the step (16-byte records cut their rounds inside the kernel); the
    # other layouts run one kernel sequence per round: price the pack calls' wall time then
    k_ms = sum(kernel_ms) / len(kernel_ms) if in_kernel else sum(pack_s) / len(pack_s) * 1e3
    achieved = algo_bytes / (k_ms * 1e-3) / 1e9
    kname = pk.tuning_name() if pk.last_algo() == "onesweep16" else pk.last_algo()
    # variant names start with the kernel family's letter (t = the plain kernel)
    kernel = pk.last_kernel()
    if kernel == "k_onesweep16":
        kernel += "<" if kname.startswith("t") else kname[:1]
    if not in_kernel:
        kernel += f" x {stats['rounds']} rounds (one kernel sequence per round; wall time of the pack calls)"
    out = {
        "metric": "device-resident KV records/s hash-partitioned & packed",
        "value": value,
        "unit": "records/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": data,
        "GBps_algorithmic": value * algo_bytes / (n * chunks) / 1e9,
        "config": {
            "workload": (what + f", hash-partition+pack into {P} destination partitions"
                         + ((", then RCCL all-to-all" if backend == "nccl" else
                             ", then a gloo exchange through host memory (rehearsal, not xGMI)")
                            if distributed else "")),
            "records_per_gpu": n * chunks, "record_bytes": total_bytes / n, "partitions": P,
            "calls_per_step": chunks,
            "local_bypass": distributed, "kernel_variant": kname,
            "partition_bytes": buf_size, "rounds": stats["rounds"], "rounds_note": rounds_note,
            "layout": "streams" if streams_layout else ("images" if comm_size else "one round"),
        },
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": committed_traffic(kernel, n, stats["rounds"]) if not distributed else None,
                     "traffic_source": ("profiles/pmc_traffic.json (committed rocprofv3 PMC passes of this "
                                        "command; not collected by this run)"
                                        if not distributed and committed_traffic(kernel, n, stats["rounds"]) else None),
                     "kernel": kernel, "kernel_ms": k_ms,
                     "algorithmic_bytes_per_launch": algo_bytes},
    }
    if distributed:
        pack_only = n * chunks * world / (sum(pack_s) / len(pack_s))
        out["pack_only_records_per_s"] = pack_only
        xfer_s = dt / args.steps - sum(pack_s) / len(pack_s)
        hot = stats["sent_max"]   # bytes of the busiest (this rank -> peer) pair over the whole step
        if split > 1:
            messages = (f"the batch is packed in {split} calls; what a call has finished travels, one message per "
                        "(source, destination) pair, while the next call packs; the rounds are views on both sides")
        elif batch_exchange:
            messages = "one per (source, destination) pair for the whole batch; the rounds are views on both sides"
        else:
            messages = "one grouped send/recv batch per round"
        out["exchange"] = {"exchanges": stats["exchanges"],
                           "messages": messages, "split": split,
                           "hottest_pair_bytes": hot,
                           "exchange_ms": xfer_s * 1e3,
                           "xgmi_floor_ms": hot / (XGMI_LINK_GBS * 1e9) * 1e3,
                           "frac_of_xgmi_pair_roofline": (hot / (XGMI_LINK_GBS * 1e9)) / max(xfer_s, 1e-9)}
        if streams_layout and "last" in stats:
            import threading
            limit = float(os.environ.get("MGS_BENCH_EXTRAS_TIMEOUT", "300"))

            def bail():
                # the measurement stands; say what is missing and leave (peers do the same)
                if rank == 0:
                    out["exchange"].update(extra)
                    out["exchange"]["post_timing"] = (f"the comparison of the exchange protocols / the verification "
                                                      f"did not finish within {limit:.0f} s and was abandoned")
                    out["cpu_baseline"] = None
                    print(json.dumps(out), flush=True)
                os._exit(0)

            dog = threading.Timer(limit, bail)
            dog.daemon = True
            dog.start()
            try:
                post_timing_extras()
            except Exception as e:  # a rank that fails leaves its peers in a collective: the watchdog ends them
                extra["post_timing_error"] = repr(e)[:400]
                print(f"# rank {rank}: {e!r}", file=sys.stderr, flush=True)
                if world > 1:
                    dog.join()      # leave through the watchdog together with the peers
            dog.cancel()
        out["exchange"].update(extra)
        if "batch_ms" in extra:
            # pack alone + the whole batch as one exchange, against the step as it ran
            serial_ms = sum(pack_s) / len(pack_s) * 1e3 + extra["batch_ms"]
            out["exchange"]["overlap_ms_hidden"] = max(0.0, serial_ms - ms_per_step) if split > 1 else 0.0
    if rank == 0 and not distributed and not args.no_cpu_baseline and args.workload == "u64":
        del images
        torch.cuda.empty_cache()
        single = cpu_baseline(records, n)
        out["cpu_baseline"] = cpu_baseline_multicore(n) or single
        out["cpu_baseline"]["single_core"] = {k: single[k] for k in ("value", "unit", "GBps", "sample")}
        out["cpu_baseline"].setdefault("host_cpus", host_cpus())
        out["cpu_baseline"]["reference_pieces"] = cpu_baseline_ref_pieces(records, n)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    pk.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
