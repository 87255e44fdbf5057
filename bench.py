#!/usr/bin/env python3
"""bench.py -- all-pairs NCD hot path on MI355X (BASELINE.json metric; lz4 by default).

One "step" = one pass of the hot path over one batch: the frame sizes of
``rows_per_step x N`` ordered genome pairs (a tile of rows of the N x N matrix
of ref:snacc/cli.py:120-129) against the N synthetic genomes resident in HBM,
followed -- when more than one rank runs -- by the RCCL all-gather of the tile.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

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
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LCG_A = 6364136223846793005
LCG_C = 1442695040888963407


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


def pmc_traffic_bytes(rows, n, length, codec="lz4"):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (profiles/r01_pmc_traffic.json:
    separate --pmc FETCH_SIZE / WRITE_SIZE passes, KB units, FETCH_SIZE doubled per the gfx950
    calibration of MI355X_MICROARCH.md).  Only valid for the launch shape it was collected on."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json" if codec == "lz4" else f"r01_pmc_traffic_{codec}.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if (t["rows"], t["genomes"], t["length"]) == (rows, n, length):
            return t["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genomes", type=int, default=1024, help="N (BASELINE configs[2]: 1024)")
    ap.add_argument("--length", type=int, default=1_000_000, help="bases per genome (configs[2]: 1 Mbp)")
    ap.add_argument("--rows-per-step", type=int, default=84, help="rows of the N x N matrix per step and rank")
    ap.add_argument("--lanes", type=int, default=0, help="override fast_lanes")
    ap.add_argument("--waves", type=int, default=0, help="override fast_waves")
    ap.add_argument("--force-generic", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="extra backend option key=value (repeatable)")
    ap.add_argument("--cpu-sample-pairs", type=int, default=0, help="pairs in the cpu_baseline sample (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--codec", choices=["lz4", "gzip", "zlib"], default="lz4")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    # SNACC_BENCH_REHEARSE=1: exercise the multi-rank code path on ONE GPU (all ranks on device 0,
    # gloo with host staging for the gather).  Never used for reported numbers.
    rehearse = os.environ.get("SNACC_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from snacc_amd.hip_backend import HipContext

    N, L, R = args.genomes, args.length, args.rows_per_step
    R = min(R, max(1, N // world))
    t0 = time.time()
    genomes = lcg_genomes_torch(N, L, 1, dev)          # seed = 1 + genome index
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
    # current stream for the RCCL all-gather, so the collective is ordered after the kernel.  (The
    # default stream's handle is 0, which the C-ABI reads as "use the context's own stream".)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    tile = torch.zeros((R, N), dtype=torch.int32, device=dev)     # u32 sizes, viewed as int32
    gathered = torch.zeros((world * R, N), dtype=torch.int32, device=dev) if world > 1 else None

    def gather():
        if world == 1:
            return
        if rehearse:                                   # gloo: stage through the host
            stream.synchronize()
            parts = [torch.zeros((R, N), dtype=torch.int32) for _ in range(world)]
            dist.all_gather(parts, tile.cpu())
            gathered.copy_(torch.cat(parts).to(dev))
        else:
            dist.all_gather_into_tensor(gathered, tile)

    kern_ms = []

    def launch(r0):
        if deflate:                                      # raw deflate stream sizes straight into the device tile
            ctx.deflate_pairs_device(args.codec, r0, r0 + R, tile.data_ptr(), stream.cuda_stream)
            kern_ms.append(ctx.deflate_last_ms())        # (waits for this launch: its event pair)
        else:
            ctx.pairs_device(r0, r0 + R, tile.data_ptr(), stream.cuda_stream)

    def step(k):
        r0 = shard0 + (k * R) % max(rows_per_rank - R + 1, 1)
        launch(r0)
        gather()
        return r0

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    fence()
    ctx.sync(stream.cuda_stream)
    kern_ms.clear()

    # HIP events on the stream the kernels are launched on (torch's current stream is passed to the
    # C-ABI).  One pair around the whole timed region: per-launch pairs under-report back-to-back
    # launches (the "start" marker of launch k+1 is stamped while launch k is still draining).
    ev_begin, ev_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fence()
    t0 = time.perf_counter()
    last_r0 = 0
    ev_begin.record(stream)
    for k in range(args.steps):
        r0 = shard0 + ((args.warmup + k) * R) % max(rows_per_rank - R + 1, 1)
        launch(r0)
        gather()
        last_r0 = r0
    ev_end.record(stream)
    fence()
    elapsed = time.perf_counter() - t0
    ctx.sync(stream.cuda_stream)                          # raises if a kernel flagged an error
    lib_last_ms = ctx.last_pairs_ms()                     # the library's own event pair, last launch

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    # Launch duration of the dominant kernel: the library brackets every launch with its own
    # hipEvent pair on the launch stream (snk_last_pairs_ms); rocprofv3 --kernel-trace agrees with it
    # (profiles/).  A torch event recorded on the idle stream before the first launch is stamped
    # late on this ROCm build, so the region pair is kept only as a cross-check.
    kern_ms_region = ev_begin.elapsed_time(ev_end) / args.steps
    kern_ms_avg = lib_last_ms if lib_last_ms > 0 else kern_ms_region
    if deflate:
        kern_ms_avg = float(np.mean(kern_ms))

    pairs_per_step = R * N * world
    pair_rate = pairs_per_step * args.steps / elapsed
    ncd_rate = pair_rate / 2.0                            # 1 NCD = 2 ordered pair-compressions
    alg_bytes_launch = R * N * (2 * L + 4)                # per launch (one rank): len_i + len_j read + 4 B written
    achieved = alg_bytes_launch / (kern_ms_avg * 1e-3) / 1e9

    gather_ok = None
    if world > 1:                                       # every rank's tile must be in its slot of the gather
        gather_ok = bool(torch.equal(gathered[rank * R:(rank + 1) * R], tile))
        flag = torch.tensor([1 if gather_ok else 0], dtype=torch.int32, device="cpu" if rehearse else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gather_ok = bool(flag.item())

    # spot parity of the last tile against the oracle (checker only; not in the timed region)
    parity = None
    cpu_baseline = None
    if rank == 0:
        import oracle
        from oracle.loader import pairs_mt
        host_tile = tile.cpu().numpy().view(np.uint32)
        js = [0, 1, N // 2, N - 1]
        if deflate:
            from oracle import deflate as dfl_oracle
            lvl = {"gzip": 9, "zlib": 6}[args.codec]
            parity = all(int(host_tile[0, j]) == dfl_oracle.raw_size(genomes[last_r0], genomes[j], lvl) for j in js[:2])
        else:
            parity = all(int(host_tile[0, j]) == oracle.lz4f_size_pair(genomes[last_r0], genomes[j]) for j in js)
        if deflate and not args.no_cpu_baseline and world == 1:
            # the codec the reference itself calls (stdlib gzip / zlib, GIL released), one thread per core
            import gzip as _gzip
            import zlib as _zlib
            fn = {"gzip": _gzip.compress, "zlib": _zlib.compress}[args.codec]
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            raw = [bytes(g) for g in genomes[:min(N, 64)]]
            budget_s = 20.0                              # bounded sample: every thread compresses pairs until the deadline
            import threading
            counts = [0] * cores
            t0 = time.perf_counter()

            def work(t):
                i = t
                while time.perf_counter() - t0 < budget_s and (not args.cpu_sample_pairs or counts[t] * cores < args.cpu_sample_pairs):
                    fn(raw[i % len(raw)] + raw[(7 * i + 1) % len(raw)])
                    counts[t] += 1
                    i += cores

            threads = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
            for th in threads:
                th.start()
            for th in threads:
                th.join()
            dt = time.perf_counter() - t0
            npairs = sum(counts)
            cpu_baseline = {
                "value": npairs / dt / 2.0, "unit": "NCD/s", "cores": cores, "kind": "reference",
                "pair_compressions_per_s": npairs / dt,
                "sample": f"{npairs} ordered pairs of {L} bp genomes, {args.codec}.compress of the interpreter "
                          f"(zlib {_zlib.ZLIB_RUNTIME_VERSION}), {cores} threads, {dt:.1f} s wall",
            }
        elif not args.no_cpu_baseline and world == 1:
            cores = os.cpu_count() or 1
            per_pair_s = 8.3e-3 * (2 * L / 2e6)           # survey probe: 8.3 ms per 2 Mbp pair per core
            want = args.cpu_sample_pairs or int(max(cores, min(20.0 / max(per_pair_s, 1e-6), 4096)))
            rows = max(1, min(N, (want + N - 1) // N))
            sub = genomes if rows * N <= want * 2 else genomes[:max(2, want // rows)]
            t0 = time.perf_counter()
            ref = pairs_mt(sub, 0, min(rows, len(sub)), cores)
            dt = time.perf_counter() - t0
            npairs = ref.size
            cpu_baseline = {
                "value": npairs / dt / 2.0, "unit": "NCD/s", "cores": cores, "kind": "port",
                "pair_compressions_per_s": npairs / dt,
                "sample": f"{npairs} ordered pairs ({ref.shape[0]} rows x {ref.shape[1]} genomes of {L} bp), "
                          f"oracle C restatement, {cores} pthreads, {dt:.1f} s wall",
            }

    if rank == 0:
        line = {
            "metric": f"genome-pair NCDs/sec ({args.codec}, all ordered pairs; 1 NCD = 2 pair-compressions)",
            "value": ncd_rate, "unit": "NCD/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic (LCG uniform ACGT, seed = 1 + genome index)",
            "config": {"workload": f"{N} synthetic {L} bp genomes, {args.codec}, rows_per_step={R} x {N} cols per GPU",
                       "genomes": N, "length": L, "rows_per_step_per_gpu": R,
                       "parallelism": f"row-shard x{world}" + (" + RCCL all-gather" if world > 1 else "")},
            "pair_compressions_per_s": pair_rate,
            "matrix_wall_s_est": (N * N + N) / pair_rate,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": pmc_traffic_bytes(R, N, L, args.codec),
                         "kernel": "dfl_parse_kernel" if deflate else ("snk_fast_kernel" if not args.force_generic else "snk_generic_kernel"),
                         "kernel_ms_avg": kern_ms_avg, "kernel_ms_region_torch_events": kern_ms_region,
                         "alg_bytes_per_launch": alg_bytes_launch},
            "cpu_baseline": cpu_baseline,
            "parity_spot_check": parity, "allgather_check": gather_ok,
            "setup_s": {"generate": round(t_gen, 2), "upload_and_singles": round(t_upload, 2)},
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
