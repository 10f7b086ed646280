#!/usr/bin/env python3
"""bench.py -- encode+decode GB/s (input bytes) of the HIP hot path, with roofline and CPU baseline.

  python bench.py [--gpus N --steps K --warmup W --workload NAME]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over this rank's batch, inputs resident in HBM:
  K1 histogram -> K2 code build -> payload offsets -> K3 encode -> [all-gather of per-chunk sizes, N > 1]
  -> K4 decode.  value = (bytes all ranks processed) / (max over ranks of the timed region).
Weak scaling: every rank holds `--bytes-per-gpu` of the stream (rank r owns the r-th contiguous chunk range).
Rank 0 prints ONE JSON line.  At N = 1 rank 0 also times the CPU oracle on a bounded sample (cpu_baseline).
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="random8g", choices=sorted(WORKLOADS))
    ap.add_argument("--bytes-per-gpu", type=int, default=0, help="override the workload's per-GPU bytes")
    ap.add_argument("--chunk-bytes", type=int, default=0)
    ap.add_argument("--cpu-sample-mib", type=int, default=-1,
                    help="MiB of the stream the CPU oracle is timed on; -1 = auto (10-30 s of CPU work), 0 = off")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; nccl (= RCCL over xGMI) is the product path, gloo only exists "
                         "to rehearse the N>1 code path on a one-GPU box")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (needs --backend gloo)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    pkg = entry.load_package()  # raises if libdczhip.so is missing: no fallback
    from dcz_amd import sharding

    gen, seed, per_gpu, chunk, desc = WORKLOADS[args.workload]
    if args.bytes_per_gpu:
        per_gpu = args.bytes_per_gpu
    if args.chunk_bytes:
        chunk = args.chunk_bytes
    per_gpu = (per_gpu // chunk) * chunk or chunk
    k_local = per_gpu // chunk
    k_total = k_local * world
    start = rank * per_gpu  # this rank's contiguous span of the stream

    svc = pkg.HipCompressionService(chunk_size_mb=max(1, chunk >> 20), device=local_rank)
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

    def step():
        svc.compress_device(t_in, chunk, out=blk)
        if world > 1:  # the one real exchange step: per-chunk compressed sizes -> global payload offsets
            sharding.gather_chunk_sizes(blk.comp_size, k_total)
        svc.decompress_device(blk.payload, blk.comp_off, blk.comp_size, orig, blk.code_lengths, chunk, t_out=t_out,
                              status=dstatus, errpos=derrpos)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)

    verified = None
    if not args.no_verify:
        step()
        torch.cuda.synchronize(dev)
        verified = bool(int(blk.status.abs().sum().item()) == 0 and int(dstatus.abs().sum().item()) == 0
                        and torch.equal(t_out, t_in))
        if not verified:
            raise SystemExit("round trip is not bit-exact on rank %d" % rank)

    svc.ctx.reset_profiling()
    svc.ctx.set_profiling(True)  # hipEvents around every kernel, on the stream the kernels run on
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    svc.ctx.set_profiling(False)

    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    comp_bytes = int(blk.total.item())
    kern = {}
    for kid, name in pkg.native.KERNEL_NAMES.items():
        ms, launches = svc.ctx.kernel_time(kid)
        kern[name] = {"ms_total": ms, "launches": launches, "avg_ms": (ms / launches) if launches else 0.0}
    # algorithmic bytes per STEP; a kernel may be launched more than once per step (the compress call is pipelined
    # in two halves), so per-launch bytes = per-step bytes * steps / launches
    alg_bytes = {"k1_histogram": per_gpu, "k3_encode": per_gpu + comp_bytes, "k4_decode": comp_bytes + per_gpu}
    # K1 fused with the identity copy (include/dcz.h DCZ_K_HISTOGRAM_COPY: every block's payload is its input, stored where
    # it belongs while it is counted): the encoder then moves N in + C out instead of 2N in + C out, K3 has nothing to move
    fused = kern["k1_histogram_copy"]["launches"] > 0 and kern["k1_histogram"]["launches"] == 0
    if fused:
        alg_bytes = {"k1_histogram_copy": per_gpu + comp_bytes, "k4_decode": comp_bytes + per_gpu}
    enc_alg = (per_gpu + comp_bytes) if fused else (2 * per_gpu + comp_bytes)
    enc_read = per_gpu if fused else 2 * per_gpu
    for name, nb in alg_bytes.items():
        k = kern[name]
        k["ms_per_step"] = k["ms_total"] / args.steps
        k["alg_bytes_per_launch"] = nb * args.steps // max(1, k["launches"])
        k["gbps"] = (k["alg_bytes_per_launch"] / (k["avg_ms"] * 1e-3) / 1e9) if k["avg_ms"] > 0 else 0.0
    dominant = max(alg_bytes, key=lambda nm: kern[nm]["ms_total"])

    if rank == 0:
        # roofline.traffic: HBM bytes of the dominant kernel family from an OFFLINE rocprofv3 --pmc pass (tools/traffic.sh ->
        # profiles/pmc_traffic.json); only quoted when that pass ran this very workload shape, otherwise null + the reason
        traffic, traffic_source = None, None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        try:
            with open(pmc_path) as f:
                ent = json.load(f).get(args.workload, {})
            if (ent.get("bytes_per_gpu"), ent.get("chunk_bytes")) != (per_gpu, chunk):
                traffic_source = "null: profiles/pmc_traffic.json has no pass for %s at %d bytes, %d-byte chunks" % (
                    args.workload, per_gpu, chunk)
            elif ent.get(dominant, {}).get("total") is None:
                traffic_source = "null: no counters for %s in profiles/pmc_traffic.json" % dominant
            else:  # the PMC figure is per step; report it per launch like `achieved`
                traffic = ent[dominant]["total"] * args.steps // max(1, kern[dominant]["launches"])
                traffic_source = "profiles/pmc_traffic.json (offline rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes, %s)" % (
                    ent.get("build", "build not recorded"))
        except Exception as e:
            traffic_source = "null: %s" % e
        value = world * per_gpu * args.steps / elapsed / 1e9
        # encode and decode separately (rank 0's own times).  t_dec = the K4 launches of a step,
        # t_enc = the rest of the step (K1/K2/K3 overlap on two streams, so their kernel times do not add up).
        step_ms = 1e3 * elapsed / args.steps
        t_dec = kern["k4_decode"]["ms_total"] / args.steps
        t_enc = max(step_ms - t_dec, 1e-9)
        split = {"t_enc_ms": round(t_enc, 4), "t_dec_ms": round(t_dec, 4),
                 "enc_alg_gbps": round(enc_alg / (t_enc * 1e-3) / 1e9, 2),
                 "dec_alg_gbps": round((per_gpu + comp_bytes) / (t_dec * 1e-3) / 1e9, 2) if t_dec > 0 else None,
                 "enc_read_gbps": round(enc_read / (t_enc * 1e-3) / 1e9, 2),
                 "enc_read_frac_of_peak": round(enc_read / (t_enc * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                 "encoder": "K1 fused with the identity copy (N in + C out)" if fused else "K1, K2, K3 (2N in + C out)"}
        line = {
            "metric": "encode+decode GB/s (input bytes)", "value": round(value, 3), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": desc, "name": args.workload, "bytes_per_gpu": per_gpu, "chunk_bytes": chunk,
                       "chunks_per_gpu": k_local, "compressed_bytes_per_gpu": comp_bytes,
                       "sharding": "contiguous chunk ranges per rank; all-gather of per-chunk sizes (RCCL)"},
            "verified_bit_exact_round_trip": verified,
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(kern[dominant]["gbps"], 2),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(kern[dominant]["gbps"] / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "traffic_source": traffic_source,
                         "alg_bytes_per_launch": kern[dominant]["alg_bytes_per_launch"],
                         "avg_launch_ms": round(kern[dominant]["avg_ms"], 4)},
            "roundtrip_roofline": {"alg_bytes_per_step": enc_alg + per_gpu + comp_bytes,
                                   "achieved": round(world * (enc_alg + per_gpu + comp_bytes) * args.steps / elapsed / 1e9, 2),
                                   "peak": HBM_PEAK_GBPS * world, "unit": "GB/s",
                                   "frac": round((enc_alg + per_gpu + comp_bytes) * args.steps / elapsed / 1e9 / HBM_PEAK_GBPS, 4)},
            "split": split,
            "kernels": {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                        for k, v in kern.items()},
        }
        # K5 beside the codec: per-chunk SHA-256 of the resident input, one lane per chunk.
        # Not part of `value` (8(d) excludes CHECKSUM_* stages).
        svc.sha256_device(t_in, chunk)
        torch.cuda.synchronize(dev)
        ts = time.perf_counter()
        for _ in range(2):
            svc.sha256_device(t_in, chunk)
        torch.cuda.synchronize(dev)
        line["sha256_per_chunk"] = {"gbps": round(2 * per_gpu / (time.perf_counter() - ts) / 1e9, 2), "chunks": k_local,
                                    "note": "one lane per chunk; throughput scales with the number of chunks"}
        if world == 1 and args.cpu_sample_mib != 0:
            line["cpu_baseline"] = cpu_baseline(np, args.workload, gen, seed, chunk, args.cpu_sample_mib)
        print(json.dumps(line), flush=True)

    svc.close()
    if world > 1:
        dist.destroy_process_group()


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
