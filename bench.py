#!/usr/bin/env python3
"""bench.py -- all-pairs NCD hot path on MI355X (BASELINE.json metric; lz4 by default).

One "step" = one pass of the hot path over one batch: the frame sizes of
``rows_per_step x N`` ordered genome pairs (a tile of rows of the N x N matrix
of ref:snacc/cli.py:120-129) against the N synthetic genomes resident in HBM,
followed -- when more than one rank runs -- by the RCCL all-gather of the tile
(issued asynchronously: the gather of step k runs under the kernel of step k+1).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

``value`` is the weak-scaling rate of the timed steps (driver contract).  After the timed region the
run also MEASURES the second half of BASELINE.json's metric, the wall time of one full N x N matrix
(``matrix_wall_s``: upload, singles + prefix snapshots, all N rows split over the ranks, the gather,
D2H, NCD assembly -- strong scaling), and times the CPU baselines.  ``--mode strong`` makes the matrix
run the reported value instead.

``--codec gzip|zlib`` measures the deflate path the same way; the default and
the headline metric is lz4.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
The oracle (oracle/) is used only by the ``cpu_baseline`` leg and a spot parity
check, never by the measured path.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LCG_A = 6364136223846793005
LCG_C = 1442695040888963407

# What actually limits the 2-bit kernel: one probe per trip of the steady loop for each of the 84
# chains the LDS of a CU holds, and a lone wave per SIMD issues one instruction per ~4.2 cycles (tools/gpu_lat.hip).
# The issue-slot floor of a trip = (instructions per wave-trip, from the committed SQ counters of this kernel) x 4.2; what
# the measured trip has on top of it is exposed waiting (L1 window, LDS / VMEM issue stalls).
ISSUE_MODEL = {"cycles_per_issue_slot": 4.2, "waitcnt_and_branch_slots_per_trip": 5, "chains_per_cu": 84, "cus": 256, "clock_hz": 2.4e9}
PROFILE_ROUNDS = ("r04", "r03", "r02")   # newest committed summary first


def lcg_genomes_torch(n_genomes, length, seed0, device):
    """SURVEY.md 8c generator, vectorised: s_i = a^i*s0 + c*(1+a+...+a^(i-1)) mod 2^64.
    int64 arithmetic wraps mod 2^64; bits 33..34 are unaffected by the arithmetic shift."""
    import torch
    a = torch.full((length,), LCG_A, dtype=torch.int64, device=device)
    apow = torch.cumprod(a, 0)                                   # a^1 .. a^L
    geo = torch.cumsum(torch.cat([torch.ones(1, dtype=torch.int64, device=device), apow[:-1]]), 0)
    cg = geo * LCG_C                                             # c * sum_{k<i} a^k
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    out = []
    for g in range(n_genomes):
        s = apow * (seed0 + g) + cg
        out.append(lut[((s >> 33) & 3)].cpu().numpy())
    return out


def _profile(stem):
    """(parsed JSON, "profiles/<file>") of the newest committed summary `<round>_<stem>.json`, or (None, None)."""
    for rnd in PROFILE_ROUNDS:
        name = f"{rnd}_{stem}.json"
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return json.load(f), f"profiles/{name}"
        except (OSError, ValueError):
            continue
    return None, None


def pmc_traffic(rows, n, length, codec="lz4", data="lcg"):
    """HBM bytes per launch from the committed rocprofv3 PMC summary of the kernel
    (profiles/rNN_pmc_traffic*.json: separate --pmc FETCH_SIZE / WRITE_SIZE passes, KB units, FETCH_SIZE
    doubled per the gfx950 calibration of MI355X_MICROARCH.md; the file names the commit it was taken at).
    Only valid for the launch shape and data set it was collected on; None otherwise."""
    stem = "pmc_traffic" if codec == "lz4" else f"pmc_traffic_{codec}"
    if data != "lcg":
        stem += f"_{data}"
    t, src = _profile(stem)
    try:
        if t and (t["rows"], t["genomes"], t["length"]) == (rows, n, length) and t.get("data", "lcg") == data:
            return t["hbm_bytes_per_launch"], f"{src} @ {t.get('collected_at_commit', '?')}"
    except KeyError:
        pass
    return None, None


def cycle_account():
    """Where the waves of the 2-bit kernel spend their cycles (stats build, tools/gpu_account.py), from the committed summary:
    the bound that limits the kernel is instruction issue and dependent latency inside its probe loop, not HBM."""
    a, src = _profile("cycle_account")
    try:
        return {"cycles_per_trip_in_loop": a["cycles_per_trip_in_loop"], "trips_per_loop_entry": a["trips_per_entry"],
                "share_outside_loop": a["share_outside_loop"], "cycles_per_exit_outside_loop": a["cycles_per_exit_outside_loop"],
                "source": f"{src} @ {a.get('collected_at_commit', '?')} (diagnostic build with the account kept in registers, "
                          f"{a['genomes']} x {a['length']} bp: a replay, not a measurement of this run)"}
    except (TypeError, KeyError):
        return None


def issue_model():
    """Issue slots of one wave-trip of the 2-bit kernel's steady loop from the committed SQ counters (tools/gpu_pmc.sh):
    VALU + SALU + LDS + VMEM instructions per wave-trip, plus the loop's s_waitcnt / branch slots."""
    q, src = _profile("pmc_sq")
    try:
        d = q["derived_per_wave_trip"]
        slots = d["valu"] + d["salu"] + d["lds"] + d["vmem"] + ISSUE_MODEL["waitcnt_and_branch_slots_per_trip"]
        return {"issue_slots_per_trip": slots, "measured_cycles_per_trip": d["cycles"],
                "probes_per_chain_trip": d.get("probes_per_chain_trip", 1.0),       # two lanes per chain: role 1's probe counts in ~2 trips of 3
                "source": f"{src} @ {q.get('collected_at_commit', '?')}"}
    except (TypeError, KeyError):
        return None


def lcg_related_torch(n_genomes, length, device, ancestors=16):
    """Secondary data set of SURVEY.md 8d: genomes = 2 % point mutants of 16 LCG ancestors (ancestor a = LCG seed 1 + a;
    genome g >= 16 = ancestor g % 16 mutated with the LCG of seed 1000 + g over the positions: substitute where
    (s >> 40) % 50 == 0 by "ACGT"[(s >> 33) & 3], the mutant rule of SURVEY.md 8c)."""
    import torch
    a = torch.full((length,), LCG_A, dtype=torch.int64, device=device)
    apow = torch.cumprod(a, 0)
    geo = torch.cumsum(torch.cat([torch.ones(1, dtype=torch.int64, device=device), apow[:-1]]), 0)
    cg = geo * LCG_C
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    anc = [lut[(((apow * (1 + k) + cg) >> 33) & 3)] for k in range(ancestors)]
    out = []
    for g in range(n_genomes):
        if g < ancestors:
            out.append(anc[g].cpu().numpy())
            continue
        s = apow * (1000 + g) + cg
        hit = (((s >> 40) & 0xFFFFFF) % 50) == 0
        out.append(torch.where(hit, lut[(s >> 33) & 3], anc[g % ancestors]).cpu().numpy())
    return out


def markov_genomes_torch(n_genomes, length, device, seed=20261004):
    """Secondary data set with the statistics uniform random DNA lacks: an order-3 Markov chain over ACGT (one fixed,
    moderately skewed transition table; 16 independently started segments per genome so that the chain runs as a wide
    vector), then tandem repeats (unit 1 .. 60 bases x 3 .. 40 copies, one per ~25 kbp) and a few point mutations inside
    them -- the low-complexity runs and long matches that drive the 2-bit kernel's rare paths."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    rng = np.random.default_rng(seed)
    cum = torch.tensor(np.cumsum(rng.dirichlet([2.0] * 4, size=64), axis=1)[:, :3], dtype=torch.float32, device=device)
    seg = 16
    steps = (length + seg - 1) // seg
    lanes = n_genomes * seg
    state = torch.randint(0, 64, (lanes,), generator=g, device=device)
    out = torch.empty((steps, lanes), dtype=torch.uint8, device=device)
    block = 2048
    for t0 in range(0, steps, block):
        u = torch.rand((min(block, steps - t0), lanes), generator=g, device=device)
        for k in range(u.shape[0]):
            nxt = (u[k].unsqueeze(1) > cum[state]).sum(1)
            out[t0 + k] = nxt.to(torch.uint8)
            state = ((state << 2) | nxt) & 63
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    seqs = lut[out.long()].view(steps, n_genomes, seg).permute(1, 2, 0).reshape(n_genomes, seg * steps)[:, :length].cpu().numpy()
    res = []
    for i in range(n_genomes):
        s = seqs[i].copy()
        for _ in range(max(1, length // 25000)):
            unit, copies = int(rng.integers(1, 61)), int(rng.integers(3, 41))
            p = int(rng.integers(0, max(1, length - unit * copies - 1)))
            n = min(unit * copies, length - p)
            s[p:p + n] = np.tile(s[p:p + unit], copies)[:n]
            if n > 40:
                s[p + int(rng.integers(0, n))] = b"ACGT"[int(rng.integers(0, 4))]
        res.append(s)
    return res


def softmask_genomes(genomes, pct, seed=11):
    """Secondary data set: the genomes with `pct` % of their bases in lower case, in stretches of 300 - 700 bases (soft-masked
    repeats as genome assemblies carry them; the rule of tools/gpu_exc.py `softPCT`).  The 2-bit kernel serves such sets with
    its other-case machinery."""
    out = []
    for g, a in enumerate(genomes):
        rng = np.random.default_rng(seed + g)
        a = a.copy()
        n = len(a)
        for s0 in rng.integers(0, max(n - 600, 1), max(1, n * pct // 100 // 500)):
            a[s0:s0 + int(rng.integers(300, 700))] |= 0x20
        out.append(a)
    return out


def secondary_leg(name, genomes, codec, local_rank, dev, R, steps, opts, data_key):
    """One short leg behind the headline region, under the same clock: upload, 1 warm-up launch, `steps` timed launches of R
    rows x N columns on a fresh context, a spot check of the last tile against the oracle, the roofline figures."""
    import torch
    from snacc_amd.hip_backend import HipContext
    N, L = len(genomes), len(genomes[0])
    deflate = codec != "lz4"
    t0 = time.perf_counter()
    ctx = HipContext(local_rank, **dict(opts, **({"defer_singles": 1} if deflate else {})))
    try:
        ctx.upload(genomes)
        if deflate:
            ctx.deflate_singles(codec)
        elif R == 84:
            R = ctx.fast_chains()                      # (a set with other-case letters holds 83 chains per CU: its rows per step follow)
        t_setup = time.perf_counter() - t0
        stream = torch.cuda.Stream(dev)
        tile = torch.zeros((R, N), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        kern_ms = []

        def launch(r0):
            if deflate:
                ctx.deflate_pairs_device(codec, r0, r0 + R, tile.data_ptr(), stream.cuda_stream)
                kern_ms.append(ctx.deflate_last_ms())
            else:
                ctx.pairs_device(r0, r0 + R, tile.data_ptr(), stream.cuda_stream)

        launch(0)
        ctx.sync(stream.cuda_stream)
        kern_ms.clear()
        if not deflate:
            ctx.pairs_ms_log()
        last_r0 = 0
        t0 = time.perf_counter()
        for k in range(steps):
            last_r0 = ((k + 1) * R) % max(N - R + 1, 1)
            launch(last_r0)
        ctx.sync(stream.cuda_stream)
        elapsed = time.perf_counter() - t0
        if not deflate:
            kern_ms = ctx.pairs_ms_log()
        host = tile.cpu().numpy().view(np.uint32)
    finally:
        ctx.close()
    import oracle
    if deflate:
        from oracle import deflate as dfl_oracle
        lvl = {"gzip": 9, "zlib": 6}[codec]
        parity = all(int(host[0, j]) == dfl_oracle.raw_size(genomes[last_r0], genomes[j], lvl) for j in (0, 1))
    else:
        parity = all(int(host[i, j]) == oracle.lz4f_size_pair(genomes[last_r0 + i], genomes[j])
                     for i, j in ((0, 0), (0, N - 1), (R - 1, 1), (R // 2, N // 2)))
    kavg = float(np.mean(kern_ms)) if len(kern_ms) else elapsed / max(steps, 1) * 1e3
    alg = R * N * (2 * L + 4)
    achieved = alg / (kavg * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic(R, N, L, codec, data_key)
    return {"pair_compressions_per_s": R * N * steps / elapsed, "ncd_per_s": R * N * steps / elapsed / 2.0,
            "kernel_ms_avg": kavg, "steps": steps, "rows_per_step": R, "setup_s": round(t_setup, 3),
            "parity_spot_check": bool(parity), "codec": codec, "data": name,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "traffic_source": traffic_src}}


def write_fasta_files(directory, genomes, width=80):
    """One single-record FASTA file per genome, 80-column lines, without a Python loop per line."""
    os.makedirs(directory, exist_ok=True)
    for i, a in enumerate(genomes):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        full = len(a) // width * width
        with open(os.path.join(directory, f"g{i:05d}.fasta"), "wb") as f:
            f.write(b">g%05d synthetic\n" % i)
            body = np.empty((full // width, width + 1), dtype=np.uint8)
            body[:, :width] = a[:full].reshape(-1, width)
            body[:, width] = 10
            f.write(body.tobytes())
            if full < len(a):
                f.write(a[full:].tobytes() + b"\n")


def cli_wall(genomes, codec, tmp_root):
    """End to end through the drop-in CLI: `snacc <dir of N FASTA files> -c <codec> -o out.csv` (FASTA ingest by the
    library's host threads, upload, singles, all N x N pairs, NCD, CSV).  The files are written before the clock starts."""
    import shutil
    from snacc_amd.cli import cli
    d = os.path.join(tmp_root, f"snacc_bench_fa_{os.getpid()}")
    try:
        write_fasta_files(d, genomes)
        out = os.path.join(d, "out.csv")
        import contextlib
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(sys.stderr):          # the CLI's banners: stdout carries the ONE JSON line only
            cli.main(args=[d, "-o", out, "-c", codec, "--no-show-progress", "--no-log"], standalone_mode=False)
        wall = time.perf_counter() - t0
        ok = os.path.getsize(out) > 0
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return {"cli_wall_s": wall, "files": len(genomes), "csv_written": ok,
            "what": f"snacc <{len(genomes)} FASTA files> -c {codec} -o out.csv --no-show-progress --no-log, in process; files written untimed"}


def cgroup_cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = float(f.read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def host_cores():
    """(threads to run = CPUs this process may use, capped by the container's CPU quota; physical cores the
    affinity mask spans)."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    phys = set()
    try:
        cur = {}
        with open("/proc/cpuinfo") as f:
            for line in f.read().split("\n") + [""]:
                if not line.strip():
                    if "processor" in cur and int(cur["processor"]) in cpus:
                        phys.add((cur.get("physical id", "0"), cur.get("core id", cur["processor"])))
                    cur = {}
                elif ":" in line:
                    k, v = line.split(":", 1)
                    cur[k.strip()] = v.strip()
    except OSError:
        pass
    threads = len(cpus)
    quota = cgroup_cpu_quota()
    if quota is not None:
        threads = max(1, min(threads, int(quota + 0.5)))
    return threads, min(threads, len(phys) or len(cpus))


def cpu_baseline_lz4(genomes, length, budget_s):
    """B2: the oracle's C restatement, one preallocated buffer + stream state per thread, for `budget_s`
    seconds on every thread this process may use.  B3 (when a liblz4 binary exists on the box): the same
    loop on LZ4F_compressFrame itself."""
    from oracle.loader import pairs_timed
    threads, phys = host_cores()
    sub = genomes[:64]
    done, dt = pairs_timed(sub, threads, budget_s)
    out = {"value": done / dt / 2.0, "unit": "NCD/s", "cores": threads, "physical_cores": phys, "kind": "port",
           "pair_compressions_per_s": done / dt,
           "sample": f"{done} ordered pairs of {length} bp genomes (drawn from the first {len(sub)}), oracle C restatement, "
                     f"{threads} pthreads with preallocated buffers, {dt:.1f} s wall"}
    try:
        from oracle import liblz4_ref
        if liblz4_ref.available():
            import ctypes
            L = liblz4_ref._load()
            raw = [bytes(g) for g in sub[:16]]
            counts = [0] * threads
            budget3 = max(2.0, budget_s / 2)
            t0 = time.perf_counter()

            def work(t):
                bufs = [raw[(t + k) % len(raw)] + raw[(7 * t + 3 * k + 1) % len(raw)] for k in range(4)]
                cap = L.LZ4F_compressFrameBound(len(bufs[0]), None)
                dst = ctypes.create_string_buffer(cap)
                k = 0
                while time.perf_counter() - t0 < budget3:
                    L.LZ4F_compressFrame(dst, cap, bufs[k & 3], len(bufs[k & 3]), None)      # ctypes drops the GIL
                    counts[t] += 1
                    k += 1

            ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            dt3 = time.perf_counter() - t0
            out["liblz4"] = {"pair_compressions_per_s": sum(counts) / dt3, "version": liblz4_ref.version(),
                             "sample": f"{sum(counts)} x LZ4F_compressFrame(2 x {length} B, NULL prefs), {threads} threads, {dt3:.1f} s"}
    except Exception as e:                                  # noqa: BLE001  (optional leg)
        out["liblz4"] = {"error": repr(e)}
    return out


def cpu_baseline_deflate(genomes, length, codec, budget_s):
    """The codec the reference itself calls (stdlib gzip / zlib, GIL released), one thread per usable core."""
    import gzip as _gzip
    import zlib as _zlib
    fn = {"gzip": _gzip.compress, "zlib": _zlib.compress}[codec]
    threads, phys = host_cores()
    raw = [bytes(g) for g in genomes[:64]]
    counts = [0] * threads
    t0 = time.perf_counter()

    def work(t):
        i = t
        while time.perf_counter() - t0 < budget_s:
            fn(raw[i % len(raw)] + raw[(7 * i + 1) % len(raw)])
            counts[t] += 1
            i += threads

    ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    dt = time.perf_counter() - t0
    npairs = sum(counts)
    return {"value": npairs / dt / 2.0, "unit": "NCD/s", "cores": threads, "physical_cores": phys, "kind": "reference",
            "pair_compressions_per_s": npairs / dt,
            "sample": f"{npairs} ordered pairs of {length} bp genomes, {codec}.compress of the interpreter "
                      f"(zlib {_zlib.ZLIB_RUNTIME_VERSION}), {threads} threads, {dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genomes", type=int, default=1024, help="N (BASELINE configs[2]: 1024)")
    ap.add_argument("--length", type=int, default=1_000_000, help="bases per genome (configs[2]: 1 Mbp)")
    ap.add_argument("--rows-per-step", type=int, default=84, help="rows of the N x N matrix per step and rank")
    ap.add_argument("--mode", choices=["weak", "strong"], default="weak",
                    help="weak (default, driver contract): fixed rows per rank per step; strong: value = the measured full-matrix run")
    ap.add_argument("--lanes", type=int, default=0, help="override fast_lanes")
    ap.add_argument("--waves", type=int, default=0, help="override fast_waves")
    ap.add_argument("--force-generic", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="extra backend option key=value (repeatable)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="wall budget of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-matrix", action="store_true", help="skip the measured full-matrix run")
    ap.add_argument("--codec", choices=["lz4", "gzip", "zlib"], default="lz4")
    ap.add_argument("--data", choices=["lcg", "related", "markov"], default="lcg",
                    help="lcg (default, the metric's data set: i.i.d. uniform ACGT); related: 2 %% mutants of 16 ancestors; "
                         "markov: order-3 Markov chain + tandem repeats (secondary data sets, SURVEY.md 8d)")
    ap.add_argument("--no-cli-wall", action="store_true", help="skip the end-to-end CLI run (FASTA files -> CSV)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short secondary legs behind the headline region (markov / related / soft-masked 5 %% data, gzip, zlib)")
    ap.add_argument("--secondary-steps", type=int, default=2)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # SNACC_BENCH_REHEARSE=1: exercise the multi-rank code path on ONE GPU (all ranks on device 0,
    # gloo with host staging for the gather).  Never used for reported numbers.
    rehearse = os.environ.get("SNACC_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0

    import torch
    import torch.distributed as dist
    from snacc_amd import distributed as sdist

    # Rendezvous first: nothing touches the GPU before the process group exists, and a failure ends the run
    # with status 3 and a message on stderr (snacc_amd.distributed.init_process_group).
    dev = torch.device("cuda", local_rank)
    if world > 1:
        sdist.init_process_group("gloo" if rehearse else "nccl", None if rehearse else dev)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    torch.cuda.set_device(local_rank)

    from snacc_amd.hip_backend import HipContext

    N, L, R = args.genomes, args.length, args.rows_per_step
    R = min(R, max(1, N // world))
    t0 = time.time()
    if args.data == "lcg":
        genomes = lcg_genomes_torch(N, L, 1, dev)      # seed = 1 + genome index
        data_desc = "synthetic (LCG uniform ACGT, seed = 1 + genome index)"
    elif args.data == "related":
        genomes = lcg_related_torch(N, L, dev)
        data_desc = "synthetic (2 % point mutants of 16 LCG ancestors; secondary data set)"
    else:
        genomes = markov_genomes_torch(N, L, dev)
        data_desc = "synthetic (order-3 Markov chain + tandem repeats; secondary data set)"
    t_gen = time.time() - t0

    opts = {}
    if args.lanes:
        opts["fast_lanes"] = args.lanes
    if args.waves:
        opts["fast_waves"] = args.waves
    if args.force_generic:
        opts["force_generic"] = 1
    for kv in args.opt:
        k, v = kv.split("=")
        opts[k] = int(v)
    ctx = HipContext(local_rank, **opts)
    t0 = time.time()
    ctx.upload(genomes)                                  # H2D + classify + pack + singles/snapshots (untimed)
    deflate = args.codec != "lz4"
    if deflate:
        ctx.deflate_singles(args.codec)                  # match index + every sequence's own symbol stream (untimed)
    t_upload = time.time() - t0

    # row shard of this rank (weak scaling: every rank does R rows per step from its own shard)
    rows_per_rank = N // world
    shard0 = rank * rows_per_rank
    # An explicit (non-default) torch stream: its handle is what the C-ABI launches on, and it is the
    # current stream when the collective is issued, so the gather is ordered after the kernel.  (The
    # default stream's handle is 0, which the C-ABI reads as "use the context's own stream".)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    # two tiles / two gather targets: the asynchronous gather of step k overlaps the kernel of step k+1
    tiles = [torch.zeros((R, N), dtype=torch.int32, device=dev) for _ in range(2)]
    gathered = [torch.zeros((world * R, N), dtype=torch.int32, device=dev) for _ in range(2)] if world > 1 else [None, None]
    pending = [None, None]

    def gather(slot):
        if world == 1:
            return
        if rehearse:                                   # gloo: stage through the host (synchronous)
            stream.synchronize()
            g, _ = sdist.gather_tile(tiles[slot].cpu(), world)
            gathered[slot].copy_(g.to(dev))
        else:
            _, pending[slot] = sdist.gather_tile(tiles[slot], world, out=gathered[slot], async_op=True)

    kern_ms = []

    def launch(r0, slot):
        if pending[slot] is not None:                    # the gather that last read this tile must be done
            pending[slot].wait()
            pending[slot] = None
        if deflate:                                      # raw deflate stream sizes straight into the device tile
            ctx.deflate_pairs_device(args.codec, r0, r0 + R, tiles[slot].data_ptr(), stream.cuda_stream)
            kern_ms.append(ctx.deflate_last_ms())        # (waits for this launch: its event pair)
        else:
            ctx.pairs_device(r0, r0 + R, tiles[slot].data_ptr(), stream.cuda_stream)

    def row0(k):
        return shard0 + (k * R) % max(rows_per_rank - R + 1, 1)

    def drain():
        for s in (0, 1):
            if pending[s] is not None:
                pending[s].wait()
                pending[s] = None

    def fence():
        drain()
        if world > 1:
            try:
                dist.barrier()
            except Exception as e:                       # noqa: BLE001
                sdist._die("barrier", e)
        torch.cuda.synchronize()

    for k in range(args.warmup):
        launch(row0(k), k & 1)
        gather(k & 1)
    fence()
    ctx.sync(stream.cuda_stream)
    kern_ms.clear()
    if not deflate:
        ctx.pairs_ms_log()                                # forget the warm-up launches

    ev_begin, ev_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fence()
    t0 = time.perf_counter()
    last_r0, last_slot = 0, 0
    ev_begin.record(stream)
    for k in range(args.steps):
        r0 = row0(args.warmup + k)
        launch(r0, k & 1)
        gather(k & 1)
        last_r0, last_slot = r0, k & 1
    ev_end.record(stream)
    fence()
    elapsed = time.perf_counter() - t0
    ctx.sync(stream.cuda_stream)                          # raises if an lz4 kernel flagged an error
    tile = tiles[last_slot]

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    # Launch duration of the dominant kernel: the library brackets EVERY launch with its own hipEvent pair on
    # the launch stream (snk_pairs_ms_log); the mean over the timed launches is reported (rocprofv3
    # --kernel-trace agrees with it, profiles/).  The torch event pair around the region is a cross-check.
    kern_ms_region = ev_begin.elapsed_time(ev_end) / max(args.steps, 1)
    if not deflate:
        kern_ms = ctx.pairs_ms_log()
    kern_ms_avg = float(np.mean(kern_ms)) if len(kern_ms) else kern_ms_region

    pairs_per_step = R * N * world
    pair_rate = pairs_per_step * args.steps / elapsed
    alg_bytes_launch = R * N * (2 * L + 4)                # per launch (one rank): len_i + len_j read + 4 B written
    achieved = alg_bytes_launch / (kern_ms_avg * 1e-3) / 1e9

    gather_ok = None
    if world > 1:                                       # every rank's tile must be in its slot of the gather
        gather_ok = sdist.allgather_check(gathered[last_slot].cpu() if rehearse else gathered[last_slot],
                                          tile.cpu() if rehearse else tile, rank, world)

    # ---- the metric's second half, MEASURED: one full N x N matrix (strong scaling over the ranks) ----
    matrix = None
    if not args.no_matrix:
        from snacc_amd.matrix import ncd_matrix_raw
        torch.cuda.set_stream(torch.cuda.default_stream(dev))
        fence()
        m0 = time.perf_counter()
        # phase A on demand for a sharded run (every rank: its own rows; the sizes ride on the tile gathers) and for gzip /
        # zlib (never needed); a single rank computes it as part of the upload
        ctx2 = HipContext(local_rank, **dict(opts, **({"defer_singles": 1} if (deflate or world > 1) else {})))
        ctx2.upload(genomes)
        stages = ctx2.upload_times()
        singles = None
        if deflate:
            singles = ctx2.deflate_singles(args.codec)
        elif world == 1:
            singles = ctx2.singles()
        m1 = time.perf_counter()
        if world == 1:
            pairs = ctx2.deflate_pairs(args.codec) if deflate else ctx2.pairs()     # (deflate sizes include the wrapper bytes)
        elif rehearse:
            pairs = None
        elif deflate:
            pairs = sdist.all_pairs_deflate_hip(ctx2, N, args.codec, lengths=ctx2.lengths())
        else:
            pairs, singles = sdist.all_pairs_hip(ctx2, N, lengths=ctx2.lengths(), with_singles=True)
        m2 = time.perf_counter()
        ncd = ncd_matrix_raw(singles, pairs) if (rank == 0 and pairs is not None) else None
        m3 = time.perf_counter()
        ctx2.close()
        wall = m3 - m0
        if world > 1:
            tmax = torch.tensor([wall], dtype=torch.float64, device="cpu" if rehearse else dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            wall = float(tmax.item())
        if pairs is not None:
            matrix = {"matrix_wall_s": wall, "upload_and_singles_s": m1 - m0, "pairs_and_gather_s": m2 - m1, "ncd_assembly_s": m3 - m2,
                      # what does not shrink with more ranks: every rank uploads all sequences (and, alone, runs phase A of all of
                      # them), rank 0 assembles the matrix.  Under --gpus N phase A is per owner and sits in pairs_and_gather_s.
                      "fixed_s": (m1 - m0) + (m3 - m2), "upload_stages_s": {k: round(v, 4) for k, v in stages.items()},
                      "strong_pair_compressions_per_s": (N * N + N) / wall, "strong_ncd_per_s": (N * N + N) / wall / 2.0,
                      "ncds": N * (N + 1) // 2, "symmetric": bool(ncd is None or np.array_equal(ncd, ncd.T)),
                      "what": "upload + singles/snapshots + all N rows (split over the ranks) + gather + D2H + float64 NCD matrix, one run"}

    # spot parity of the last tile against the oracle (checker only; not in the timed region)
    parity = None
    cpu_baseline = None
    probes_per_pair = None
    if rank == 0:
        import oracle
        host_tile = tile.cpu().numpy().view(np.uint32)
        js = [0, 1, N // 2, N - 1]
        if deflate:
            from oracle import deflate as dfl_oracle
            lvl = {"gzip": 9, "zlib": 6}[args.codec]
            parity = all(int(host_tile[0, j]) == dfl_oracle.raw_size(genomes[last_r0], genomes[j], lvl) for j in js[:2])
        else:
            parity = all(int(host_tile[0, j]) == oracle.lz4f_size_pair(genomes[last_r0], genomes[j]) for j in js)
            # probes of one pair job (the oracle's statistics): those of x+y minus those of the blocks of x that
            # come from its prefix snapshot
            from oracle.loader import lz4f_size_stats
            xy = np.concatenate([genomes[0], genomes[1]])
            _, st_xy = lz4f_size_stats(xy)
            _, st_x = lz4f_size_stats(genomes[0])
            covered = (L // 65536) * 65536 / max(L, 1) if L > 65536 else 0.0
            probes_per_pair = (st_xy["search_probes"] + st_xy["chain_probes"]) - covered * (st_x["search_probes"] + st_x["chain_probes"])
        if not args.no_cpu_baseline and world == 1:
            cpu_baseline = (cpu_baseline_deflate(genomes, L, args.codec, args.cpu_seconds) if deflate
                            else cpu_baseline_lz4(genomes, L, args.cpu_seconds))
    cli_run = None
    if rank == 0 and world == 1 and not args.no_cli_wall:
        try:
            cli_run = cli_wall(genomes, args.codec, os.environ.get("TMPDIR", "/tmp"))
        except Exception as e:                              # noqa: BLE001  (an extra of the line: e.g. no room for the FASTA files)
            cli_run = {"cli_wall_s": None, "error": repr(e)}

    # ---- secondary legs, under the same clock as the headline (after its timed region; the headline keys are untouched) ----
    secondary = None
    if rank == 0 and world == 1 and not args.no_secondary and args.codec == "lz4" and args.data == "lcg":
        secondary = {}
        ctx.close()                                        # the headline context's arenas: room for the legs' own
        legs = [("markov", lambda: markov_genomes_torch(N, L, dev), "lz4", "markov"),
                ("related", lambda: lcg_related_torch(N, L, dev), "lz4", "related"),
                ("softmask5", lambda: softmask_genomes(genomes, 5), "lz4", "softmask5"),
                ("gzip", lambda: genomes, "gzip", "lcg"),
                ("zlib", lambda: genomes, "zlib", "lcg")]
        for name, make, codec, key in legs:
            t_leg = time.perf_counter()
            try:
                gs = make()
                t_made = time.perf_counter() - t_leg
                secondary[name] = secondary_leg(name, gs, codec, local_rank, dev, R, args.secondary_steps, opts, key)
                secondary[name]["generate_s"] = round(t_made, 2)
                del gs
            except Exception as e:                         # noqa: BLE001  (an extra of the line)
                secondary[name] = {"error": repr(e)}
            secondary[name]["leg_wall_s"] = round(time.perf_counter() - t_leg, 2)

    if rank == 0:
        if args.mode == "strong" and matrix:
            pair_rate = (N * N + N) / matrix["matrix_wall_s"]
        ncd_rate = pair_rate / 2.0                        # 1 NCD = 2 ordered pair-compressions
        traffic, traffic_src = pmc_traffic(R, N, L, args.codec, args.data)
        issue_bound = None
        im = issue_model() if probes_per_pair else None
        if im:
            mdl = ISSUE_MODEL
            floor = im["issue_slots_per_trip"] * mdl["cycles_per_issue_slot"]
            peak = mdl["cus"] * mdl["chains_per_cu"] * mdl["clock_hz"] / floor * im["probes_per_chain_trip"]
            ach = R * N * probes_per_pair / (kern_ms_avg * 1e-3)
            issue_bound = {"bound": "instruction issue of one wave per SIMD x chains resident in LDS (the limiter; HBM is idle)",
                           "achieved": ach, "peak": peak, "unit": "probes/s per GPU", "frac": ach / peak,
                           "probes_per_pair": probes_per_pair,
                           "model": dict(mdl, issue_slots_per_trip=im["issue_slots_per_trip"], issue_floor_cycles_per_trip=floor,
                                         probes_per_chain_trip=im["probes_per_chain_trip"],
                                         measured_cycles_per_trip=im["measured_cycles_per_trip"],
                                         exposed_wait_cycles_per_trip=im["measured_cycles_per_trip"] - floor, source=im["source"])}
        par = f"row-shard x{world}"
        if world > 1:
            par += " + gloo all-gather (ONE-GPU REHEARSAL, not a measurement)" if rehearse else " + RCCL all-gather (async, overlapped)"
        line = {
            "metric": f"genome-pair NCDs/sec ({args.codec}, all ordered pairs; 1 NCD = 2 pair-compressions)",
            "value": ncd_rate, "unit": "NCD/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
            "higher_is_better": True, "scaling": args.mode, "vs_baseline": None,
            "dtype": "u8", "data": data_desc,
            "config": {"workload": f"{N} synthetic {L} bp genomes, {args.codec}, rows_per_step={R} x {N} cols per GPU",
                       "genomes": N, "length": L, "rows_per_step_per_gpu": R, "parallelism": par},
            "pair_compressions_per_s": pair_rate,
            "matrix_wall_s": matrix["matrix_wall_s"] if matrix else None,
            "matrix": matrix,
            "matrix_wall_s_est": (N * N + N) / (pairs_per_step * args.steps / elapsed),
            # achieved / frac: ALGORITHMIC bytes per launch over the launch time; traffic: HBM bytes the
            # counters saw; hbm_gbps / hbm_frac: that traffic over the same time -- what HBM itself is asked for
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                         "hbm_gbps": (traffic / (kern_ms_avg * 1e-3) / 1e9) if traffic else None,
                         "hbm_frac": (traffic / (kern_ms_avg * 1e-3) / 1e9 / 8000.0) if traffic else None,
                         "kernel": "dfl_parse_kernel" if deflate else ("snk_fast_kernel" if not args.force_generic else "snk_bytes_compact_kernel"),
                         "kernel_ms_avg": kern_ms_avg, "kernel_launches_averaged": len(kern_ms),
                         "kernel_ms_region_torch_events": kern_ms_region,
                         "alg_bytes_per_launch": alg_bytes_launch},
            "issue_bound": issue_bound,
            "cli_wall_s": cli_run["cli_wall_s"] if cli_run else None, "cli": cli_run,
            "cycle_account": cycle_account() if args.codec == "lz4" else None,
            "cpu_baseline": cpu_baseline,
            "parity_spot_check": parity, "allgather_check": gather_ok,
            "secondary": secondary,
            # the strong-scaling figure beside the weak one (the driver's --gpus N runs read `value`; this is the whole matrix, one run)
            "strong": ({"pair_compressions_per_s": matrix["strong_pair_compressions_per_s"], "ncd_per_s": matrix["strong_ncd_per_s"],
                        "matrix_wall_s": matrix["matrix_wall_s"], "fixed_s": matrix["fixed_s"], "n_gpus": world} if matrix else None),
            "setup_s": {"generate": round(t_gen, 2), "upload_and_singles": round(t_upload, 2)},
        }
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
