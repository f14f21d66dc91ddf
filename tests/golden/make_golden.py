"""Regenerates the golden vectors in this directory from the REFERENCE ITSELF.

Runs only where /root/reference exists: oracle/Makefile compiles the
reference's own src/hash.cpp and the bytestream<T> templates of
src/serializer.h into oracle/_ref/libmimir_ref.so, and this script records
their outputs on seeded inputs.  The JSON files hold inputs and expected
outputs only (data, no reference source).

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle.oracle import Reference, ReferenceSerializer  # noqa: E402


def main():
    ref = Reference()
    rng = np.random.default_rng(20251111)

    # 1. hashlittle(key, len, 0) for every length 0..64 (four keys each) and a
    #    few long keys, each hashed at byte misalignments 0..3 so that all
    #    three load strategies of src/hash.cpp (:166, :286, :344) are recorded.
    vectors = []
    lengths = [l for l in range(0, 65) for _ in range(4)] + [
        100, 255, 256, 257, 1000, 1023, 1024, 4097]
    for L in lengths:
        key = rng.integers(0, 256, size=L, dtype=np.uint8).tobytes()
        hs = [ref.hashlittle(key, 0, misalign=a) for a in range(4)]
        assert len(set(hs)) == 1, "reference hash depends on alignment?"
        vectors.append({"key_hex": key.hex(), "len": L, "hash": hs[0]})
    # printable keys in the shapes the examples emit
    for word in [b"a", b"b", b"zz", b"hello", b"world", b"mimir", b"x" * 1023,
                 b"Four score and seven years ago"]:
        key = word + b"\0"
        vectors.append({"key_hex": key.hex(), "len": len(key),
                        "hash": ref.hashlittle(key)})
    for v in [0, 1, 2, 0xdeadbeefcafef00d, 0xffffffffffffffff]:
        key = int(v).to_bytes(8, "little")
        vectors.append({"key_hex": key.hex(), "len": 8,
                        "hash": ref.hashlittle(key)})
    octkey = b"012345670123456"
    for L in range(1, 16):
        vectors.append({"key_hex": octkey[:L].hex(), "len": L,
                        "hash": ref.hashlittle(octkey[:L])})
    # initval != 0 (not used by the shuffle path, kept for completeness)
    seeded = []
    for L in (0, 5, 12, 13, 31):
        key = rng.integers(0, 256, size=L, dtype=np.uint8).tobytes()
        for iv in (1, 0x9e3779b9):
            seeded.append({"key_hex": key.hex(), "len": L, "initval": iv,
                           "hash": ref.hashlittle(key, iv)})
    with open(os.path.join(HERE, "hashlittle_vectors.json"), "w") as f:
        json.dump({"source": "src/hash.cpp compiled by oracle/Makefile",
                   "initval0": vectors, "seeded": seeded}, f, indent=0)

    # 2. wire forms written by bytestream<T>::to_bytes for the four record
    #    shapes of the examples, including the "-1 when it does not fit" rule.
    wire = []
    for word in [b"a", b"hello", b"Zq9", b"w" * 40]:
        for val in (1, 0x0102030405060708):
            out = ref.kv_cstr_u64(word, val)
            wire.append({"shape": "cstr_u64", "key": word.decode(), "val": val,
                         "bytes_hex": out.hex()})
        need = len(word) + 9
        wire.append({"shape": "cstr_u64_fit", "key": word.decode(), "val": 1,
                     "cap_exact_ok": ref.kv_cstr_u64(word, 1, cap=need) is not None,
                     "cap_minus1_ok": ref.kv_cstr_u64(word, 1, cap=need - 1) is not None})
    for k, v in [(0, 0), (1, 2), (0xdeadbeefcafef00d, 7), (2**64 - 1, 2**63)]:
        wire.append({"shape": "u64_u64", "key": k, "val": v,
                     "bytes_hex": ref.kv_u64_u64(k, v).hex()})
    for level in (1, 4, 8, 12, 15):
        wire.append({"shape": "chars_u64", "key": octkey.decode(),
                     "keycount": level, "val": 1,
                     "bytes_hex": ref.kv_chars_u64(octkey, level, 1).hex()})
    val16 = bytes([0xaa, 0, 0, 0, 0, 0, 0, 0]) + (1234).to_bytes(8, "little")
    for word in [b"k", b"joinkey"]:
        wire.append({"shape": "cstr_blob16", "key": word.decode(),
                     "val_hex": val16.hex(),
                     "bytes_hex": ref.kv_cstr_blob16(word, val16).hex()})
    with open(os.path.join(HERE, "wire_vectors.json"), "w") as f:
        json.dump({"source": "bytestream<T>::to_bytes, src/serializer.h:40-223",
                   "vectors": wire}, f, indent=0)
    # 3. wordcount's map over small texts: FileParser::to_line framing, the reader's
    #    strlen + 1 step, strtok_r split, bytestream serialisation.
    texts = [b"", b"\n", b" ", b"a", b"a\n", b"hello world\n", b"  two  spaces \n\n end",
             b"tab\tstays cr\r\n x", b"nul\0splits the line\nnext", b"\0\0 \n\0a",
             b"x" * 1023 + b" " + b"y" * 1024 + b" z\n", b"w" * 2500, b"a b\n" * 40,
             bytes(range(1, 256)) + b"\n"]
    tok = []
    for t in texts:
        recs, offs = ref.wordcount_map(np.frombuffer(t, dtype=np.uint8), 1)
        tok.append({"text_hex": t.hex(), "records_hex": bytes(recs).hex(),
                    "rec_off": [int(x) for x in offs]})
    with open(os.path.join(HERE, "tokenize_vectors.json"), "w") as f:
        json.dump({"source": "FileParser::to_line (src/fileparser.h:21-41) + the reader's step "
                             "(src/filereader.h:152-171) + strtok_r split and <word,1> write as in "
                             "examples/wordcount.cpp:66-82, serialised by bytestream<T>::to_bytes",
                   "vectors": tok}, f, indent=0)
    # 4. receive side: record boundaries found by the reference's own from_bytes /
    #    kv_from_bytes walk over small receive streams (save_data).
    from mimir_amd import datagen
    parse = []
    b, _, _ = datagen.u64_records(7, seed=3)
    parse.append({"shape": "u64_u64", "arg": 0, "stream_hex": b.tobytes().hex()})
    for level in (1, 8, 15):
        b, _, _ = datagen.octree_records(9, level=level, seed=4)
        parse.append({"shape": "chars_u64", "arg": level, "stream_hex": b.tobytes().hex()})
    b, _, _ = datagen.word_records(12, n_unique=5, alpha=0.5, len_mean=5, seed=5)
    parse.append({"shape": "cstr_fixed", "arg": 8, "stream_hex": b.tobytes().hex()})
    # values full of zero bytes, empty words (a lone NUL), a 300-byte word
    hand = b"".join(w + b"\0" + (v).to_bytes(8, "little")
                    for w, v in [(b"a", 0), (b"", 0), (b"bc", 1 << 56), (b"x" * 300, 0), (b"", 256)])
    parse.append({"shape": "cstr_fixed", "arg": 8, "stream_hex": hand.hex()})
    b, _, _ = datagen.join_records(10, n_unique=4, len_mean=6, len_sd=1, seed=6)
    parse.append({"shape": "cstr_fixed", "arg": 16, "stream_hex": b.tobytes().hex()})
    parse.append({"shape": "cstr_fixed", "arg": 0, "stream_hex": b"one\0two\0\0three\0".hex()})
    parse.append({"shape": "cstr_fixed", "arg": 8, "stream_hex": ""})
    for v in parse:
        stream = np.frombuffer(bytes.fromhex(v["stream_hex"]), dtype=np.uint8)
        v["rec_off"] = [int(x) for x in ref.parse_records(v["shape"], stream, v["arg"])]
    with open(os.path.join(HERE, "parse_vectors.json"), "w") as f:
        json.dump({"source": "bytestream<T>::from_bytes / Serializer::kv_from_bytes walked as save_data does "
                             "(src/serializer.h:54-61, :100-129, :376-416; src/collectiveshuffler.h:563-585)",
                   "vectors": parse}, f, indent=0)
    print(f"wrote {len(parse)} parse vectors")
    # 5. the pack side's two Serializer methods themselves: kv_to_bytes (return value and the
    #    bytes written at buffer sizes around the key's and the record's length: the "-1 iff
    #    it does not fit" rule that ends a round) and get_hash_code (what get_target_rank
    #    takes modulo the rank count), for the record shapes of the examples.
    ser = ReferenceSerializer()
    val16 = bytes([0xbb, 0, 0, 0, 0, 0, 0, 0]) + (77).to_bytes(8, "little")
    cases = [("cstr_u64", b"a", 1, 1), ("cstr_u64", b"hello", 0x0102030405060708, 1),
             ("cstr_u64", b"w" * 40, 7, 1), ("cstr_u64", (b"left", b"right"), 3, 2),
             ("cstr_u64", (b"", b"x"), 3, 2),
             ("u64_u64", 0, 0, 1), ("u64_u64", 0xdeadbeefcafef00d, 5, 1),
             ("chars_u64", octkey, 1, 1), ("chars_u64", octkey, 1, 4), ("chars_u64", octkey, 1, 8),
             ("chars_u64", octkey, 1, 12), ("chars_u64", octkey, 1, 15),
             ("cstr_blob16", b"k", val16, 1), ("cstr_blob16", b"joinkey", val16, 1)]
    pack = []
    for shape, key, val, kc in cases:
        rc_full, rec = ser.kv_to_bytes(shape, key, val, 4096, keycount=kc)
        R = len(rec)
        assert rc_full == R
        vb = 16 if shape == "cstr_blob16" else 8
        klen = R - vb
        fits = []
        for bufsize in sorted({0, max(klen - 1, 0), klen, klen + 1, R - 1, R, R + 1}):
            rc, wrote = ser.kv_to_bytes(shape, key, val, bufsize, keycount=kc)
            fits.append({"bufsize": bufsize, "ret": rc, "written_hex": wrote.hex()})
        def enc(x):
            if isinstance(x, tuple):
                return [e.hex() for e in x]
            return x.hex() if isinstance(x, (bytes, bytearray)) else x
        pack.append({"shape": shape, "key": enc(key), "val": enc(val), "keycount": kc,
                     "record_hex": rec.hex(), "key_bytes": klen,
                     "hash_code": ser.hash_code(shape, key, keycount=kc), "fits": fits})
    with open(os.path.join(HERE, "pack_vectors.json"), "w") as f:
        json.dump({"source": "Serializer<K,V>::kv_to_bytes (src/serializer.h:348-364) and get_hash_code "
                             "(:464-480), called through oracle/ref_probe_ser.cpp",
                   "vectors": pack}, f, indent=0)
    print(f"wrote {len(pack)} pack vectors")
    print(f"wrote {len(vectors)}+{len(seeded)} hash vectors, {len(wire)} wire vectors, {len(tok)} texts")


if __name__ == "__main__":
    main()
