"""Python statement (slow, small inputs only) of the hash-table scheme of the 2-bit lz4 kernel
(snacc_amd/csrc/snk_fast.hip.h): per slot a 16-bit offset inside the current 64 KiB block plus a "written in this
block" bit; at every block transition entries without the bit are dropped (they are too far for good) and all bits
are cleared; an entry without the bit refers to the previous block and is live only if its offset is larger than the
cursor's.  Test infrastructure: the model is run through a plain LZ4-frame linked-block parse and must give the
sizes of the oracle's 4096 x u32 table (tests/test_oracle_properties.py)."""

M64 = (1 << 64) - 1


def hash5(a, p):
    v = int.from_bytes(bytes(a[p:p + 5]), "little")
    return ((((v << 24) & M64) * 889523592379) & M64) >> 52


class U16BitmapTable:
    def __init__(self):
        self.off, self.bit, self.base = {}, set(range(4096)), 0      # an empty slot is "offset 0, written": stream position 0

    def start_block(self, base, first):
        if not first:
            for h in list(self.off):
                if h not in self.bit:
                    self.off[h] = 0
            self.bit = set()
        self.base = base

    def get(self, h, cur):
        e, c = self.off.get(h, 0), cur - self.base
        if h in self.bit:
            return self.base + e, True
        return self.base - 65536 + e, e > c

    def put(self, h, pos):
        self.off[h] = pos - self.base
        self.bit.add(h)


class PlainTable:
    """liblz4's own table: absolute positions, candidates more than 65535 back are rejected."""
    def __init__(self):
        self.t = {}

    def start_block(self, base, first):
        pass

    def get(self, h, cur):
        c = self.t.get(h, 0)
        return c, c + 65535 >= cur

    def put(self, h, pos):
        self.t[h] = pos


def frame_size(a, T):
    """LZ4 frame size (linked 64 KiB blocks, n > 65536) of the byte array `a` with table model T."""
    n, total, pos, first = len(a), 7, 0, True
    while pos < n:
        blen = min(65536, n - pos)
        iend = pos + blen
        T.start_block(pos, first)
        first = False
        if blen < 13:
            total += 4 + blen
            pos = iend
            continue
        mfl1, mlimit, olimit = iend - 11, iend - 5, blen - 1
        T.put(hash5(a, pos), pos)
        ip, anchor, op, bail, done = pos + 1, pos, 0, False, False
        while not done:
            fip, step, nb = ip, 1, 64
            while True:
                cur = fip
                cand, ok = T.get(hash5(a, cur), cur)
                ip = fip
                fip += step
                step = nb >> 6
                nb += 1
                if fip > mfl1:
                    done = True
                    break
                T.put(hash5(a, cur), cur)
                if ok and bytes(a[cand:cand + 4]) == bytes(a[ip:ip + 4]):
                    break
            if done:
                break
            while ip > anchor and cand > 0 and a[ip - 1] == a[cand - 1]:
                ip -= 1
                cand -= 1
            lit = ip - anchor
            op += 1
            if op + lit + 8 + lit // 255 > olimit:
                bail = True
                break
            op += lit + ((lit - 15) // 255 + 1 if lit >= 15 else 0)
            while True:
                op += 2
                x, y = ip + 4, cand + 4
                while x < mlimit and a[x] == a[y]:
                    x += 1
                    y += 1
                mc = x - ip - 4
                ip = x
                if op + 6 + (mc + 240) // 255 > olimit:
                    bail = True
                    break
                if mc >= 15:
                    op += (mc - 15) // 255 + 1
                anchor = ip
                if ip >= mfl1:
                    done = True
                    break
                T.put(hash5(a, ip - 2), ip - 2)
                cand, ok = T.get(hash5(a, ip), ip)
                T.put(hash5(a, ip), ip)
                if ok and bytes(a[cand:cand + 4]) == bytes(a[ip:ip + 4]):
                    op += 1
                    continue
                break
            if bail or done:
                break
            ip += 1
        if bail:
            payload = blen
        else:
            run = iend - anchor
            payload = blen if op + run + 1 + (run + 240) // 255 > olimit else op + 1 + run + ((run - 15) // 255 + 1 if run >= 15 else 0)
        total += 4 + payload
        pos = iend
    return total + 4
