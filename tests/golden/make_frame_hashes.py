#!/usr/bin/env python3
"""Generate tests/golden/frame_hashes.json: sha256 + length of the LZ4 frames the liblz4 1.9.3
*binary* of this image produces for generator-defined inputs -- with NULL preferences (what the
build fixes as the serial reference) and with the frame's content-size field
switched on (the one documented degree of freedom of a real py-lz4framed wheel).

The GPU box may not have liblz4: the `-m gpu` test compares the frames `snk_frames_list` emits
(ref:snacc/pairwise_ncd.py:80-88) with these hashes unconditionally.

    python tests/golden/make_frame_hashes.py      # needs /opt/conda/lib/liblz4.so.1.9.3
"""
import ctypes
import hashlib
import json
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parents[1]))
sys.path.insert(0, str(HERE.parent))
from oracle import liblz4_ref  # noqa: E402
from frame_items import ITEMS, build_sequences  # noqa: E402


class FrameInfo(ctypes.Structure):          # lz4frame.h 1.9.3: LZ4F_frameInfo_t
    _fields_ = [("blockSizeID", ctypes.c_int), ("blockMode", ctypes.c_int), ("contentChecksumFlag", ctypes.c_int),
                ("frameType", ctypes.c_int), ("contentSize", ctypes.c_ulonglong), ("dictID", ctypes.c_uint),
                ("blockChecksumFlag", ctypes.c_int)]


class Preferences(ctypes.Structure):        # LZ4F_preferences_t
    _fields_ = [("frameInfo", FrameInfo), ("compressionLevel", ctypes.c_int), ("autoFlush", ctypes.c_uint),
                ("favorDecSpeed", ctypes.c_uint), ("reserved", ctypes.c_uint * 3)]


def compress_frame_content_size(data):
    L = liblz4_ref._load()
    prefs = Preferences()
    prefs.frameInfo.contentSize = max(len(data), 1)      # any non-zero value: liblz4 replaces it by srcSize
    n = len(data)
    cap = L.LZ4F_compressFrameBound(n, ctypes.byref(prefs))
    dst = ctypes.create_string_buffer(cap)
    r = L.LZ4F_compressFrame(dst, cap, data if n else None, n, ctypes.byref(prefs))
    assert not L.LZ4F_isError(r)
    return dst.raw[:r]


def main():
    assert liblz4_ref.available() and liblz4_ref.version() == "1.9.3"
    assert ctypes.sizeof(FrameInfo) == 32 and ctypes.sizeof(Preferences) == 56
    seqs = build_sequences()
    out = {"liblz4_version": liblz4_ref.version(), "generator": "tests/golden/make_frame_hashes.py",
           "inputs": "tests/frame_items.py", "frames": []}
    for i, j in ITEMS:
        data = seqs[i] + (seqs[j] if j >= 0 else b"")
        f0 = liblz4_ref.compress_frame(data)
        f1 = compress_frame_content_size(data)
        # a frame with the content-size field is the plain frame with an 8-byte longer header (n = 0: liblz4 drops the field)
        assert len(f1) == len(f0) + (8 if len(data) else 0) and f1[-(len(f0) - 7):] == f0[7:]
        out["frames"].append({"item": [i, j], "n": len(data),
                              "len": len(f0), "sha256": hashlib.sha256(f0).hexdigest(),
                              "len_content_size": len(f1), "sha256_content_size": hashlib.sha256(f1).hexdigest(),
                              "header_content_size_hex": f1[:15].hex() if len(data) else f1[:7].hex()})
    (HERE / "frame_hashes.json").write_text(json.dumps(out, indent=1) + "\n")
    print("wrote", len(out["frames"]), "frames")


if __name__ == "__main__":
    main()
