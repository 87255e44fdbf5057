"""Minimal LZ4 *frame* decoder (test infrastructure): the encode -> decode round trip is a
size-independent property check for the frames emitted by the GPU."""


def decode_block(src, out):
    """Decode one LZ4 block appended to bytearray `out` (which holds the linked-block history)."""
    i, n = 0, len(src)
    while i < n:
        token = src[i]; i += 1
        lit = token >> 4
        if lit == 15:
            while True:
                b = src[i]; i += 1
                lit += b
                if b != 255:
                    break
        out += src[i:i + lit]; i += lit
        if i >= n:
            break
        off = src[i] | (src[i + 1] << 8); i += 2
        ml = token & 15
        if ml == 15:
            while True:
                b = src[i]; i += 1
                ml += b
                if b != 255:
                    break
        ml += 4
        assert 0 < off <= len(out), "offset outside the window"
        start = len(out) - off
        for k in range(ml):                       # may overlap
            out.append(out[start + k])
    return out


def decode_frame(frame):
    assert frame[:4] == bytes.fromhex("04224d18"), "magic"
    flg, bd = frame[4], frame[5]
    assert flg >> 6 == 1 and bd == 0x40 and not (flg & 0x1C), "unexpected FLG/BD"
    independent = bool(flg & 0x20)
    i = 7
    out = bytearray()
    while True:
        word = int.from_bytes(frame[i:i + 4], "little"); i += 4
        if word == 0:
            break
        size = word & 0x7FFFFFFF
        data = frame[i:i + size]; i += size
        if word & 0x80000000:
            out += data
        elif independent:
            out += decode_block(data, bytearray())
        else:
            decode_block(data, out)
    assert i == len(frame), "trailing bytes"
    return bytes(out)
