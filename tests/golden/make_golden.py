"""Regenerates the golden fixtures in this directory.  Run from the repo root:
    python tests/golden/make_golden.py

Inputs are never stored (they come from tests/gen.py = the reference's generators); only
expected outputs are:
  blake3_kat.json   public BLAKE3 known answers: "", "abc", and the official test-vector input
                    pattern (byte i = i % 251) at the official lengths.  These are published
                    constants of the BLAKE3 project, typed in — NOT produced by our code — and
                    this script refuses to write the file if oracle/blake3_ref.c disagrees.
  zstd_frames.json  Zstandard frames produced by the container's libzstd 1.4.8 (an independent
                    implementation of RFC 8878) for the reference's generators, hex encoded,
                    with the BLAKE3 of the expected decoded bytes.
The reference itself ships no golden blobs/archives and cannot be built or
imported here (Rust; no cargo), so there are no reference-generated vectors.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import gen  # noqa: E402
from oracle import oracle as O  # noqa: E402

BLAKE3_KAT = {
    "empty": "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262",
    "abc": "6437b3ac38465133ffb63b75273a8db548c558465d79db03fd359c6cd5bd9d85",
    "pattern251": {
        "0": "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262",
        "1": "2d3adedff11b61f14c886e35afa036736dcd87a74d27b5c1510225d0f592e213",
        "1023": "10108970eeda3eb932baac1428c7a2163b0e924c9a9e25b35bba72b28f70bd11",
        "1024": "42214739f095a406f3fc83deb889744ac00df831c10daa55189b5d121c855af7",
        "1025": "d00278ae47eb27b34faecf67b4fe263f82d5412916c1ffd97c8cb7fb814b8444",
        "2048": "e776b6028c7cd22a4d0ba182a8bf62205d2ef576467e838ed6f2529b85fba24a",
        "2049": "5f4d72f40d7a5f82b15ca2b2e44b1de3c2ef86c426c95c1af0b6879522563030",
        "3072": "b98cb0ff3623be03326b373de6b9095218513e64f1ee2edd2525c7ad1e5cffd2",
        "3073": "7124b49501012f81cc7f11ca069ec9226cecb8a2c850cfe644e327d22d3e1cd3",
        "4096": "015094013f57a5277b59d8475c0501042c0b642e531b0a1c8f58d2163229e969",
        "4097": "9b4052b38f1c5fc8b1f9ff7ac7b27cd242487b3d890d15c96a1c25b8aa0fb995",
        "5120": "9cadc15fed8b5d854562b26a9536d9707cadeda9b143978f319ab34230535833",
        "8192": "aae792484c8efe4f19e2ca7d371d8c467ffb10748d8a5a1ae579948f718a2a63",
        "31744": "62b6960e1a44bcc1eb1a611a8d6235b6b4b78f32e7abc4fb4c6cdcce94895c47",
    },
}


def main():
    assert O.blake3(b"").hex() == BLAKE3_KAT["empty"]
    assert O.blake3(b"abc").hex() == BLAKE3_KAT["abc"]
    for n, h in BLAKE3_KAT["pattern251"].items():
        assert O.blake3(gen.binary(int(n))).hex() == h, n
    json.dump(BLAKE3_KAT, open(os.path.join(HERE, "blake3_kat.json"), "w"), indent=1)

    frames = []
    cases = [
        ("text", 10240, 19), ("text", 10240, 3), ("binary", 10240, 19), ("random_lcg", 4096, 19),
        ("text", 0, 19), ("text", 1, 19), ("text", 44, 19), ("text", 300000, 19),
        ("pseudo_text", 6000, 19), ("pseudo_text", 6000, 1), ("pseudo_text", 20000, 3), ("zeros", 70000, 19),
    ]
    for gname, n, lvl in cases:
        data = bytes(n) if gname == "zeros" else getattr(gen, gname)(n)
        f = O.libzstd_compress(data, lvl)
        assert O.zstd_decompress(f) == data
        frames.append(dict(gen=gname, size=n, level=lvl, frame_hex=f.hex(), blake3=O.blake3(data).hex()))
    json.dump(dict(producer="libzstd 1.4.8 (container), ZSTD_compressCCtx", frames=frames),
              open(os.path.join(HERE, "zstd_frames.json"), "w"), indent=1)
    print("wrote blake3_kat.json, zstd_frames.json:", sum(len(x["frame_hex"]) // 2 for x in frames), "frame bytes")


if __name__ == "__main__":
    main()
