#!/usr/bin/env python3
"""Debugging aid: runs the byte-granular one-sweep kernel in its checked build
(MGS_OSD_CHECKED=1: every store is tested against the caller's buffers; a store
outside them is counted and skipped) on a few variable-width inputs and prints
the counters, then compares with the oracle."""
import os
import sys

os.environ["MGS_OSD_CHECKED"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mimir_amd import capi, datagen  # noqa: E402
from mimir_amd.packer import DevicePacker  # noqa: E402
from oracle import oracle as O  # noqa: E402

orc = O.Oracle()
cases = [("words n=25000 buf=65536", datagen.word_records(25000, n_unique=5000, alpha=1.0, len_mean=8, len_sd=2.5, seed=4), 8, 2, 65536),
         ("words n=25000 buf=64", datagen.word_records(25000, n_unique=5000, alpha=1.0, len_mean=8, len_sd=2.5, seed=4), 8, 2, 64),
         ("words n=3e5 one round", datagen.word_records(300000, n_unique=5000, alpha=1.0, len_mean=6, len_sd=1, seed=5), 8, -1, 1 << 22)]
for name, (blob, offs, lay), P, rank, buf in cases:
    n = len(offs) - 1
    pk = DevicePacker(capi.Layout.make(*lay), P, rank, buf, n)
    image = torch.full((P * buf,), 0xEE, dtype=torch.uint8, device="cuda:0")
    local = torch.full((len(blob),), 0xDD, dtype=torch.uint8, device="cuda:0")
    d_off = torch.from_numpy(offs.astype(np.uint32).view(np.int32)).cuda()
    try:
        got = pk.pack(torch.from_numpy(blob).cuda(), n, image, rec_off=d_off, local_out=local if rank >= 0 else None)
    except capi.MgsError as e:
        print(name, "ERROR", e)
        continue
    st = pk.debug_stats()
    want = orc.pack_round(O.Layout.make(*lay), blob, offs, n, 0, P, rank, buf)
    img = image.cpu().numpy()
    same = got.next == want["next"] and got.send_offset.tolist() == want["send_offset"].tolist() and all(
        np.array_equal(img[p * buf:p * buf + int(want["send_offset"][p])], want["partitions"][p]) for p in range(P))
    print(name, "algo", pk.last_algo(), "bad stores", st[0], "last bad addr", hex(st[1]), "what/len", st[2] >> 32, st[2] & 0xffffffff,
          "block/thread", st[3] >> 32, st[3] & 0xffffffff, "image", hex(image.data_ptr()), "local", hex(local.data_ptr()),
          "equal to oracle:", same, "next", got.next, want["next"])
    pk.close()
