"""Debug aid: where does the full-size membership check differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mimir_amd import capi
from mimir_amd.packer import DevicePacker, synth_u64_records

n, P = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000, 8
rec = synth_u64_records(n, seed=42)
glay = capi.Layout.make(capi.KEY_FIXED, 8, capi.VAL_FIXED, 8)
buf = 16 * (n // P + n // 64 + 4096)
pk = DevicePacker(glay, P, -1, buf, max_records=n)
image = pk.new_send_image()
dest = torch.empty(n, dtype=torch.uint8, device="cuda:0")
got = pk.pack(rec, n, image, dest_out=dest)
print("algo", pk.last_algo(), "next", got.next, got.send_offset.tolist())
words = rec.view(torch.int64).view(n, 2)
for p in range(P):
    cnt = int(got.send_offset[p]) // 16
    idx = torch.nonzero(dest == p).squeeze(1)
    part = image[p * buf:p * buf + 16 * cnt].view(torch.int64).view(cnt, 2)
    # the value column is the emit index: compare with idx directly (no big gather)
    same_val = bool((part[:, 1] == idx).all().item())
    gathered = words[idx]
    neq = (gathered != part).any(1)
    nbad = int(neq.sum().item())
    print(f"p={p} cnt={cnt} idx={idx.numel()} values==idx {same_val} gather mismatches {nbad}")
    if nbad:
        j = int(torch.nonzero(neq)[0].item())
        print("  first at", j, "idx", int(idx[j]), "gathered", gathered[j].tolist(), "part", part[j].tolist(),
              "direct", words[int(idx[j])].tolist())
    # keys via a chunked gather
    ok = True
    for s in range(0, cnt, 1 << 24):
        e = min(cnt, s + (1 << 24))
        ok &= bool((words[:, 0][idx[s:e]] == part[s:e, 0]).all().item())
    print("  keys equal with chunked gather:", ok)
    del idx, gathered, neq
