#!/usr/bin/env python3
"""What a wrong launch-shape hint costs: a context whose completed calls were all text is handed a
deep queue of 8 GiB all-identity calls (every call gets the persistent k3_copy_identity / k4_fixed and the unfused K1),
then, after one synchronisation, the same queue again (flat k4_fixed, K1 fused with the copy).  Prints kernel times of
both; run under `rocprofv3 --kernel-trace --stats` for the per-kernel averages."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

import torch  # noqa: E402

pkg = entry.load_package()
dev = torch.device("cuda", 0)
per_gpu, chunk, steps = 8 << 30, 1 << 20, int(sys.argv[1]) if len(sys.argv) > 1 else 6
svc = pkg.HipCompressionService(1, 0)
lib, h = pkg.lib(), svc.ctx.handle
t_in = torch.empty(per_gpu, dtype=torch.uint8, device=dev)
t_txt = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
svc.ctx.check(lib.dczu_fill_java_random(h, t_in.data_ptr(), per_gpu, 42, 0, None))
svc.ctx.check(lib.dczu_fill_text(h, t_txt.data_ptr(), t_txt.numel(), 0xD0C2, 0, None))
svc.ctx.check(lib.dcz_ctx_reserve(h, per_gpu, chunk))
k = per_gpu // chunk
orig = torch.full((k,), chunk, dtype=torch.int32, device=dev)
t_out = torch.empty(per_gpu, dtype=torch.uint8, device=dev)
st, ep = torch.zeros(k, dtype=torch.int32, device=dev), torch.zeros(k, dtype=torch.int64, device=dev)
blk = svc.compress_device(t_in, chunk)
torch.cuda.synchronize()
# completed text calls: the state that makes the hint say "no identity / fixed-length blocks"
kt = t_txt.numel() // chunk
otx = torch.full((kt,), chunk, dtype=torch.int32, device=dev)
for _ in range(12):
    b2 = svc.compress_device(t_txt, chunk)
    svc.decompress_device(b2.payload, b2.comp_off, b2.comp_size, otx, b2.code_lengths, chunk)
    torch.cuda.synchronize()


def queue(tag):
    svc.ctx.reset_profiling()
    svc.ctx.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        svc.compress_device(t_in, chunk, out=blk)
        svc.decompress_device(blk.payload, blk.comp_off, blk.comp_size, orig, blk.code_lengths, chunk, t_out=t_out, status=st,
                              errpos=ep)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    svc.ctx.set_profiling(False)
    assert torch.equal(t_out, t_in) and int(st.abs().sum()) == 0
    kern = {n: svc.ctx.kernel_time(i) for i, n in pkg.native.KERNEL_NAMES.items()}
    print(json.dumps({"case": tag, "ms_per_step": round(1e3 * el / steps, 3), "gbps": round(per_gpu * steps / el / 1e9, 1),
                      "launch_shapes": svc.ctx.launch_shapes(),
                      "avg_ms": {n: round(ms / max(1, c), 4) for n, (ms, c) in kern.items()},
                      "launches": {n: c for n, (ms, c) in kern.items()}}), flush=True)


queue("hint wrong: persistent shapes with work, unfused K1")
queue("hint right: flat k4_fixed, K1 fused with the copy")
svc.close()
