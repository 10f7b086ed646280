#!/usr/bin/env python3
"""bench.py -- encode+decode GB/s (input bytes) of the HIP hot path, with roofline and CPU baseline.

  python bench.py [--gpus N --steps K --warmup W --workload NAME]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over this rank's batch, inputs resident in HBM:
  K1 histogram -> K2 code build -> payload offsets -> K3 encode -> [all-gather of per-chunk sizes, N > 1]
  -> K4 decode.  value = (bytes all ranks processed) / (max over ranks of the timed region).
Weak scaling: every rank holds `--bytes-per-gpu` of the stream (rank r owns the r-th contiguous chunk range).
Rank 0 prints ONE JSON line.  The headline fields describe `--workload` (default: the north-star target); the other
BASELINE.json configurations are timed after it with a few steps each and reported under "secondary" of the same line,
each with its own roofline.  At N = 1 rank 0 also times the CPU oracle on a bounded sample (cpu_baseline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"

WORKLOADS = {
    # name: (generator, seed, bytes per GPU, chunk bytes, description)
    "random8g": ("java_random", 42, 8 << 30, 1 << 20,
                 "8GiB/GPU uniform-random bytes (java.util.Random(42) stream), 1MiB chunks [north_star target]"),
    "random256m": ("java_random", 42, 256 << 20, 1 << 20,
                   "256MiB uniform-random bytes (java.util.Random(42)), 1MiB chunks [BASELINE config 3]"),
    "text": ("text", 0xD0C2, 1 << 30, 4 << 20,
             "1GiB/GPU order-0 English-like text (8GiB over 8 GPUs), 4MiB chunks [BASELINE config 4]"),
    "lowentropy": ("lowentropy", 0xD0C5, 8 << 30, 4 << 20,
                   "8GiB/GPU zeros + 1% noise (64GiB over 8 GPUs), 4MiB chunks [BASELINE config 5]"),
    # the reference's own chunk sizes (cli/DataCompCLI.java:35: 32 MB; application.conf:10: 16 MB): few large chunks,
    # decoded by many workgroups per chunk (k4_split.hip)
    "text_32m": ("text", 0xD0C2, 1 << 30, 32 << 20,
                 "1GiB/GPU order-0 English-like text, 32MiB chunks (the reference CLI's default chunk size)"),
    "text8g": ("text", 0xD0C2, 8 << 30, 4 << 20,
               "8GiB/GPU order-0 English-like text, 4MiB chunks [BASELINE config 4 at one GPU's worth of 8 GiB]"),
}
# what the default run times after the headline workload (name, steps, warm-up)
SECONDARY_1GPU = [("text8g", 5, 3), ("lowentropy", 5, 3), ("random256m", 20, 5), ("text", 10, 3), ("text_32m", 10, 3)]
SECONDARY_NGPU = [("text", 10, 3), ("lowentropy", 5, 3)]  # configs 4 and 5 exactly, at N = 8


class Env:
    """Process-wide pieces every workload shares."""

    def __init__(self, args):
        import numpy as np
        import torch
        import torch.distributed as dist
        self.np, self.torch, self.dist, self.args = np, torch, dist, args
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                             % (args.gpus, self.world, args.gpus))
        if args.single_device:
            self.local_rank = 0
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)  # nccl == RCCL on ROCm
            else:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
        self.pkg = entry.load_package()  # raises if libdczhip.so is missing: no fallback
        from dcz_amd import sharding
        self.sharding = sharding


def run_workload(env, name, steps, warmup, verify, per_gpu_override=0, chunk_override=0):
    """Times `steps` steps of one workload on a context of its own (the launch-shape hints are per context, so a
    workload starts like a fresh process would).  Returns the result dict of rank 0's view, value aggregated over ranks."""
    torch, dist, pkg, world, rank, dev = env.torch, env.dist, env.pkg, env.world, env.rank, env.dev
    gen, seed, per_gpu, chunk, desc = WORKLOADS[name]
    if per_gpu_override:
        per_gpu = per_gpu_override
    if chunk_override:
        chunk = chunk_override
    per_gpu = (per_gpu // chunk) * chunk or chunk
    k_local = per_gpu // chunk
    k_total = k_local * world
    start = rank * per_gpu  # this rank's contiguous span of the stream

    svc = pkg.HipCompressionService(chunk_size_mb=max(1, chunk >> 20), device=env.local_rank)
    lib, h = pkg.lib(), svc.ctx.handle
    t_in = torch.empty(per_gpu, dtype=torch.uint8, device=dev)
    fill = {"java_random": lambda: lib.dczu_fill_java_random(h, t_in.data_ptr(), per_gpu, seed, start, None),
            "text": lambda: lib.dczu_fill_text(h, t_in.data_ptr(), per_gpu, seed, start, None),
            "lowentropy": lambda: lib.dczu_fill_lowentropy(h, t_in.data_ptr(), per_gpu, seed, start, None)}[gen]
    svc.ctx.check(fill())
    torch.cuda.synchronize(dev)
    svc.ctx.check(lib.dcz_ctx_reserve(h, per_gpu, chunk))

    blk = svc.compress_device(t_in, chunk)  # allocates the output tensors once; reused by every step
    orig = torch.full((k_local,), chunk, dtype=torch.int32, device=dev)
    t_out = torch.empty(per_gpu, dtype=torch.uint8, device=dev)
    dstatus = torch.zeros(k_local, dtype=torch.int32, device=dev)
    derrpos = torch.zeros(k_local, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)

    # The caller's stream: the steps run on one torch stream whose handle goes to the library as the C ABI's `stream`
    # argument (kernels, the per-kernel hipEvents and -- at N > 1 -- the RCCL all-gather are all queued on it, in order).
    # --service-stream: the Python mirror's default instead (its own stream, ordered with the caller's by two events per call).
    ts = None if env.args.service_stream else torch.cuda.Stream(dev)
    sh = None if ts is None else ts.cuda_stream

    def step():
        with torch.cuda.stream(ts):
            svc.compress_device(t_in, chunk, out=blk, stream=sh)
            if world > 1:  # the one real exchange step: per-chunk compressed sizes -> global payload offsets
                env.sharding.gather_chunk_sizes(blk.comp_size, k_total)
            svc.decompress_device(blk.payload, blk.comp_off, blk.comp_size, orig, blk.code_lengths, chunk, t_out=t_out,
                                  status=dstatus, errpos=derrpos, stream=sh)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)

    verified = None
    if verify:
        step()
        torch.cuda.synchronize(dev)
        verified = bool(int(blk.status.abs().sum().item()) == 0 and int(dstatus.abs().sum().item()) == 0
                        and torch.equal(t_out, t_in))
        if not verified:
            raise SystemExit("round trip of %s is not bit-exact on rank %d" % (name, rank))

    svc.ctx.reset_profiling()
    svc.ctx.set_profiling(True)  # hipEvents around every kernel, on the stream the kernels run on
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):  # all steps are queued back to back: nothing in the loop waits for the device
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    svc.ctx.set_profiling(False)
    shapes = svc.ctx.launch_shapes()  # (of the timed steps: read before anything else calls into the library)

    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if env.args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # Small configurations are launch-bound (a decode call is ~11 launches, most of which find nothing to do): the same step
    # captured once in a hipGraph and replayed, reported beside the eager figure, never instead of it.
    graph = None
    if world == 1 and k_local < 2048:
        try:
            gs, g = torch.cuda.Stream(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=gs):
                svc.compress_device(t_in, chunk, out=blk, stream=gs.cuda_stream)
                svc.decompress_device(blk.payload, blk.comp_off, blk.comp_size, orig, blk.code_lengths, chunk, t_out=t_out,
                                      status=dstatus, errpos=derrpos, stream=gs.cuda_stream)
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize(dev)
            tg = time.perf_counter()
            for _ in range(steps):
                g.replay()
            torch.cuda.synchronize(dev)
            tg = time.perf_counter() - tg
            ok = bool(int(blk.status.abs().sum().item()) == 0 and int(dstatus.abs().sum().item()) == 0 and torch.equal(t_out, t_in))
            graph = {"value": round(per_gpu * steps / tg / 1e9, 3), "unit": "GB/s", "ms_per_step": round(1e3 * tg / steps, 4),
                     "steps": steps, "verified_bit_exact_round_trip": ok,
                     "note": "compress + decompress captured once in a hipGraph (launch shapes as the warm-up left them) and replayed"}
            del g
        except Exception as e:  # a capture problem must not take the bench line down
            graph = {"error": "%s: %s" % (type(e).__name__, e)}

    comp_bytes = int(blk.total.item())
    kern = {}
    for kid, kname in pkg.native.KERNEL_NAMES.items():
        ms, launches = svc.ctx.kernel_time(kid)
        kern[kname] = {"ms_total": ms, "launches": launches, "avg_ms": (ms / launches) if launches else 0.0}
    res = None
    if rank == 0:
        res = _account(env, name, desc, steps, warmup, per_gpu, chunk, k_local, comp_bytes, elapsed, kern, shapes, verified)
        # K5 beside the codec: per-chunk SHA-256 of the resident input, one lane per chunk.
        # Not part of `value` (8(d) excludes CHECKSUM_* stages).
        svc.sha256_device(t_in, chunk)
        torch.cuda.synchronize(dev)
        ts = time.perf_counter()
        for _ in range(2):
            svc.sha256_device(t_in, chunk)
        torch.cuda.synchronize(dev)
        if graph is not None:
            res["graph_replay"] = graph
        res["sha256_per_chunk"] = {"gbps": round(2 * per_gpu / (time.perf_counter() - ts) / 1e9, 2), "chunks": k_local,
                                   "note": "one lane per chunk; throughput scales with the number of chunks"}
    svc.close()
    del t_in, t_out, blk
    torch.cuda.empty_cache()
    return res


def _account(env, name, desc, steps, warmup, per_gpu, chunk, k_local, comp_bytes, elapsed, kern, shapes, verified):
    """Algorithmic bytes PER LAUNCH (K1 N read, K3 N read + C written, K4 C read + N written) from the
    bytes a launch really covers: a compress call runs as one launch of each kernel, or as two pipelined halves
    (>= 2048 chunks) -- read off K2, which launches once per range whatever else happens.  K1 exists in two variants
    (include/dcz.h DCZ_K_HISTOGRAM_COPY: it also stores the payload of identity blocks, N read + C written, and K3 then
    has nothing to move); a run in which both ran is flagged `mixed_launch_shapes` and K3 gets no figure."""
    world = env.world
    N, Cb = per_gpu, comp_bytes
    calls = steps
    lpc = max(1, round(kern["k2_codebuild"]["launches"] / calls))  # launches per compress call (1, or 2 halves)
    n_k1, n_k1c = kern["k1_histogram"]["launches"], kern["k1_histogram_copy"]["launches"]
    fused_frac = n_k1c / max(1, n_k1 + n_k1c)
    mixed = 0 < n_k1c and 0 < n_k1
    per_launch = {"k1_histogram": N / lpc if n_k1 else None,
                  "k1_histogram_copy": (N + Cb) / lpc if n_k1c else None,
                  "k3_encode": (N + Cb) / lpc if n_k1c == 0 else None,
                  "k4_decode": float(Cb + N)}
    for kname, nb in per_launch.items():
        k = kern[kname]
        k["ms_per_step"] = k["ms_total"] / steps
        if nb is None or not k["launches"] or k["avg_ms"] <= 0:
            k["alg_bytes_per_launch"], k["gbps"] = None, None
            continue
        k["alg_bytes_per_launch"] = int(nb)
        k["gbps"] = nb / (k["avg_ms"] * 1e-3) / 1e9
        if k["gbps"] > HBM_PEAK_GBPS:  # an accounting error, never a result
            raise SystemExit("bench.py accounting error: %s at %.0f GB/s exceeds the HBM peak (%d launches, %.4f ms avg)"
                             % (kname, k["gbps"], k["launches"], k["avg_ms"]))
    cand = [kn for kn in per_launch if kern[kn]["gbps"] is not None]
    dominant = max(cand, key=lambda nm: kern[nm]["ms_total"])
    enc_alg = fused_frac * (N + Cb) + (1 - fused_frac) * (2 * N + Cb)
    enc_read = fused_frac * N + (1 - fused_frac) * 2 * N

    # roofline.traffic: HBM bytes of the dominant kernel family from an OFFLINE rocprofv3 --pmc pass (tools/traffic.sh ->
    # profiles/pmc_traffic.json); quoted only when that pass ran this workload shape WITH THE SAME LAUNCH MIX (launches of
    # every kernel family per step), otherwise null + the reason
    traffic, traffic_source = None, None
    lps = {kn: kern[kn]["launches"] / steps for kn in kern}
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            ent = json.load(f).get(name, {})
        want = ent.get("launches_per_step")
        if (ent.get("bytes_per_gpu"), ent.get("chunk_bytes")) != (per_gpu, chunk):
            traffic_source = "null: profiles/pmc_traffic.json has no pass for %s at %d bytes, %d-byte chunks" % (
                name, per_gpu, chunk)
        elif want is None or any(abs(want.get(kn, 0) - lps[kn]) > 1e-9 for kn in lps):
            traffic_source = "null: the PMC pass ran another launch mix (%s) than this run (%s)" % (want, lps)
        elif ent.get(dominant, {}).get("total") is None:
            traffic_source = "null: no counters for %s in profiles/pmc_traffic.json" % dominant
        else:  # the PMC figure is per step; report it per launch like `achieved`
            traffic = int(ent[dominant]["total"] / max(1e-9, lps[dominant]))
            traffic_source = "profiles/pmc_traffic.json (offline rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes, %s)" % (
                ent.get("build", "build not recorded"))
    except Exception as e:
        traffic_source = "null: %s" % e
    value = world * per_gpu * steps / elapsed / 1e9
    # encode and decode separately (rank 0's own times).  t_dec = the K4 launches of a step,
    # t_enc = the rest of the step (K1/K2/K3 overlap on two streams, so their kernel times do not add up).
    step_ms = 1e3 * elapsed / steps
    t_dec = kern["k4_decode"]["ms_total"] / steps
    t_enc = max(step_ms - t_dec, 1e-9)
    split = {"t_enc_ms": round(t_enc, 4), "t_dec_ms": round(t_dec, 4),
             "enc_alg_gbps": round(enc_alg / (t_enc * 1e-3) / 1e9, 2),
             "dec_alg_gbps": round((per_gpu + comp_bytes) / (t_dec * 1e-3) / 1e9, 2) if t_dec > 0 else None,
             "enc_read_gbps": round(enc_read / (t_enc * 1e-3) / 1e9, 2),
             "enc_read_frac_of_peak": round(enc_read / (t_enc * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
             "encoder": ("mixed launch shapes: %.0f%% of the K1 launches fused with the identity copy" % (100 * fused_frac)) if mixed
             else "K1 fused with the identity copy (N in + C out)" if n_k1c else "K1, K2, K3 (2N in + C out)"}
    rt_alg = enc_alg + per_gpu + comp_bytes
    return {
        "value": round(value, 3), "unit": "GB/s", "steps": steps, "warmup": warmup,
        "ms_per_step": round(step_ms, 4),
        "config": {"workload": desc, "name": name, "bytes_per_gpu": per_gpu, "chunk_bytes": chunk,
                   "chunks_per_gpu": k_local, "compressed_bytes_per_gpu": comp_bytes,
                   "sharding": "contiguous chunk ranges per rank; all-gather of per-chunk sizes (RCCL)"},
        "verified_bit_exact_round_trip": verified,
        "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(kern[dominant]["gbps"], 2),
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(kern[dominant]["gbps"] / HBM_PEAK_GBPS, 4), "traffic": traffic,
                     "traffic_source": traffic_source,
                     "alg_bytes_per_launch": kern[dominant]["alg_bytes_per_launch"],
                     "avg_launch_ms": round(kern[dominant]["avg_ms"], 4)},
        "roundtrip_roofline": {"alg_bytes_per_step": int(rt_alg),
                               "achieved": round(world * rt_alg * steps / elapsed / 1e9, 2),
                               "peak": HBM_PEAK_GBPS * world, "unit": "GB/s",
                               "frac": round(rt_alg * steps / elapsed / 1e9 / HBM_PEAK_GBPS, 4)},
        "split": split,
        "launch_shapes": shapes, "mixed_launch_shapes": bool(mixed),
        "kernels": {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                    for k, v in kern.items()},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="random8g", choices=sorted(WORKLOADS))
    ap.add_argument("--bytes-per-gpu", type=int, default=0, help="override the workload's per-GPU bytes")
    ap.add_argument("--chunk-bytes", type=int, default=0)
    ap.add_argument("--cpu-sample-mib", type=int, default=-1,
                    help="MiB of the stream the CPU oracle is timed on; -1 = auto (10-30 s of CPU work), 0 = off")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="only the headline workload (profiling passes; the default run also times the other configs)")
    ap.add_argument("--service-stream", action="store_true",
                    help="launch on the Python service's own stream (two cross-stream events per call) instead of handing "
                         "the caller's stream to the C ABI")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; nccl (= RCCL over xGMI) is the product path, gloo only exists "
                         "to rehearse the N>1 code path on a one-GPU box")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (needs --backend gloo)")
    args = ap.parse_args()
    env = Env(args)

    head = run_workload(env, args.workload, args.steps, args.warmup, not args.no_verify, args.bytes_per_gpu,
                        args.chunk_bytes)
    secondary = {}
    if not args.no_secondary and not args.bytes_per_gpu and not args.chunk_bytes:
        for name, st, wu in (SECONDARY_1GPU if env.world == 1 else SECONDARY_NGPU):
            if name == args.workload:
                continue
            r = run_workload(env, name, st, wu, not args.no_verify)
            if env.rank == 0:
                secondary[name] = r

    if env.rank == 0:
        line = {"metric": "encode+decode GB/s (input bytes)", "value": head["value"], "unit": "GB/s",
                "n_gpus": env.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic"}
        for k in ("config", "verified_bit_exact_round_trip", "roofline", "roundtrip_roofline", "split", "launch_shapes",
                  "mixed_launch_shapes", "kernels", "graph_replay", "sha256_per_chunk"):
            if k in head:
                line[k] = head[k]
        if secondary:
            line["secondary"] = secondary
        if env.world == 1 and args.cpu_sample_mib != 0:
            gen, seed, _, chunk, _ = WORKLOADS[args.workload]
            line["cpu_baseline"] = cpu_baseline(env.np, args.workload, gen, seed, args.chunk_bytes or chunk,
                                                args.cpu_sample_mib)
        print(json.dumps(line), flush=True)

    if env.world > 1:
        env.dist.destroy_process_group()


def cpu_baseline(np, workload, gen, seed, chunk, sample_mib):
    """The CPU oracle (a C port of the reference's CPU path; there is no JVM to run the reference itself)
    on a bounded sample of the same workload, with the reference's worker count max(2, min(nproc, 8))
    (CpuCompressionService.java:42-44).  sample_mib < 0: a 128 MiB probe sizes the sample for ~12 s of
    CPU work (capped at 8 GiB).  Reported baseline only."""
    orc = entry.load_oracle()
    ncpu = os.cpu_count() or 1
    threads = max(2, min(ncpu, 8))
    make = {"java_random": lambda n: orc.java_random_bytes(seed, n), "text": lambda n: orc.gen_text(seed, 0, n),
            "lowentropy": lambda n: orc.gen_lowentropy(seed, 0, n)}[gen]
    if sample_mib < 0:
        probe = (128 << 20) // chunk * chunk or chunk
        e0, d0, _ = orc.roundtrip_blocks_mt(make(probe), chunk, threads)
        sample_mib = int(min(8192, max(128, 12.0 / max(e0 + d0, 1e-3) * 128)))
    n = (sample_mib << 20) // chunk * chunk or chunk
    data = make(n)
    enc_s, dec_s, comp = orc.roundtrip_blocks_mt(data, chunk, threads)
    return {"value": round(n / (enc_s + dec_s) / 1e9, 4), "unit": "GB/s", "cores": threads, "kind": "port",
            "sample": "%d MiB of the same stream, %d-byte chunks, %d chunk workers (host has %d cpus); "
                      "encode %.2fs + decode %.2fs; C restatement of CpuCompressionService (no JVM on the box)"
                      % (n >> 20, chunk, threads, ncpu, enc_s, dec_s),
            "encode_gbps": round(n / enc_s / 1e9, 4), "decode_gbps": round(n / dec_s / 1e9, 4)}


if __name__ == "__main__":
    main()
