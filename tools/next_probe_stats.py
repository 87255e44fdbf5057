"""Dev aid (CPU): how predictable is the next probe of liblz4's parse on the bench data?  Parses x+y of two LCG genomes with a
Python restatement of the block loop and counts, over the probes inside y, the distance to the next probe: the input to the
"speculative lanes" estimate in DESIGN.md section 11.  Usage: python tools/next_probe_stats.py"""
import sys, numpy as np
sys.path.insert(0, '.')
import oracle
from collections import Counter
L = 300000
x = bytes(oracle.lcg_genome(1, L)); y = bytes(oracle.lcg_genome(2, L))
base = x + y
n = len(base)
def h5(p):
    v = int.from_bytes(base[p:p+8].ljust(8, b'\0'), 'little')
    return (((v << 24) & 0xFFFFFFFFFFFFFFFF) * 889523592379 & 0xFFFFFFFFFFFFFFFF) >> 52
tab = [0] * 4096
adv = Counter(); kinds = Counter()
probes = []   # (pos, matched, matchend)
pos = 0
while pos < n:
    blen = min(65536, n - pos); iend = pos + blen
    if blen < 13: break
    mfl1 = iend - 11; mlimit = iend - 5
    ip = pos; anchor = pos
    tab[h5(ip)] = ip; ip += 1
    done = False
    while not done:
        fip = ip; step = 1; nb = 64
        while True:
            cur = fip; h = h5(cur); cand = tab[h]; ip = fip; fip += step; step = nb >> 6; nb += 1
            if fip > mfl1: done = True; break
            tab[h] = cur
            if cand + 65535 < cur: probes.append((cur, False, 0)); continue
            if base[cand:cand+4] == base[ip:ip+4]: break
            probes.append((cur, False, 0))
        if done: break
        p0 = ip
        while ip > anchor and cand > 0 and base[ip-1] == base[cand-1]: ip -= 1; cand -= 1
        while True:
            a = ip + 4; b = cand + 4
            while a < mlimit and base[a] == base[b]: a += 1; b += 1
            probes.append((p0, True, a))
            ip = a; anchor = ip
            if ip >= mfl1: done = True; break
            tab[h5(ip-2)] = ip - 2
            h = h5(ip); cand = tab[h]; tab[h] = ip
            if cand + 65535 >= ip and base[cand:cand+4] == base[ip:ip+4]:
                p0 = ip; continue
            probes.append((ip, False, 0))
            break
        if done: break
        ip += 1
    pos = iend
# statistics over probes within y
ps = [p for p in probes if p[0] >= L]
tot = len(ps)
d = Counter()
for (p, m, e), (p2, m2, e2) in zip(ps, ps[1:]):
    d[(m, p2 - p)] += 1
print("probes in y", tot, "per base", tot / L)
for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:14]:
    print(k, v, f"{v / tot:.3f}")
# chain success of speculation: predict next = p+5 given match
run = Counter(); cur_run = 0
hit1 = sum(1 for (p, m, e), (p2, _, _) in zip(ps, ps[1:]) if m and p2 - p == 5)
print("P(next probe = cur+5 after a match)", hit1 / tot)
# depth-2: two successive +5
hit2 = sum(1 for a, b, c in zip(ps, ps[1:], ps[2:]) if a[1] and b[0] - a[0] == 5 and b[1] and c[0] - b[0] == 5)
print("P(two in a row)", hit2 / tot)
# expected probes per trip with greedy grouping (depth 3, fixed +5 predictions)
i = 0; trips = 0
while i < len(ps) - 3:
    k = 1
    if ps[i][1] and ps[i+1][0] - ps[i][0] == 5:
        k = 2
        if ps[i+1][1] and ps[i+2][0] - ps[i+1][0] == 5: k = 3
    i += k; trips += 1
print("probes per trip, depth 3:", len(ps) / trips)
i = 0; trips = 0
while i < len(ps) - 3:
    k = 2 if (ps[i][1] and ps[i+1][0] - ps[i][0] == 5) else 1
    i += k; trips += 1
print("probes per trip, depth 2:", len(ps) / trips)
# round 4: three lanes per chain -- which second speculation pays more?  (a) role 2 at cur+10 (counts when two 5-base matches
# follow each other), (b) role 2 at cur+6 (counts when role 0's match ends at cur+6 and role 1's probe does not count),
# (c) both (four lanes).  Greedy grouping as above: a trip consumes the probes that counted.
def per_trip(use10, use6):
    i = 0; trips = 0
    while i < len(ps) - 3:
        k = 1
        d1 = ps[i + 1][0] - ps[i][0]
        if ps[i][1] and d1 == 5:
            k = 2
            if use10 and ps[i + 1][1] and ps[i + 2][0] - ps[i + 1][0] == 5: k = 3
        elif use6 and ps[i][1] and d1 == 6:
            k = 2
        i += k; trips += 1
    return len(ps) / trips
print("probes per trip: lanes at +5: %.3f; +5,+10: %.3f; +5,+6: %.3f; +5,+6,+10: %.3f" % (per_trip(False, False), per_trip(True, False), per_trip(False, True), per_trip(True, True)))
# ... and when the speculation after a NON-match (next probe at cur+1: a search probe, nothing owed) is served too
def per_trip2(use10, use6, use1):
    i = 0; trips = 0
    while i < len(ps) - 3:
        k = 1
        d1 = ps[i + 1][0] - ps[i][0]
        if ps[i][1] and d1 == 5:
            k = 2
            if use10 and ps[i + 1][1] and ps[i + 2][0] - ps[i + 1][0] == 5: k = 3
        elif use6 and ps[i][1] and d1 == 6:
            k = 2
        elif use1 and (not ps[i][1]) and d1 == 1:
            k = 2
        i += k; trips += 1
    return len(ps) / trips
print("with a lane at +1 as well: +5,+1: %.3f; +5,+6,+1: %.3f; +5,+10,+1: %.3f" % (per_trip2(False, False, True), per_trip2(False, True, True), per_trip2(True, False, True)))
