# Encoder: does K3's second read of the input come from the 256 MiB Infinity Cache when the compress call is cut into
# batches that fit it?    Times dcz_compress_blocks over 8 GiB as a sequence of batch-sized calls.
import sys, time, torch
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
svc = pkg.HipCompressionService(1, 0)
lib, h = pkg.lib(), svc.ctx.handle
kind = sys.argv[1] if len(sys.argv) > 1 else "random"
n, bb = 8 << 30, 1 << 20
t = torch.empty(n, dtype=torch.uint8, device="cuda")
{"random": lambda: lib.dczu_fill_java_random(h, t.data_ptr(), n, 42, 0, None),
 "text": lambda: lib.dczu_fill_text(h, t.data_ptr(), n, 0xD0C2, 0, None)}[kind]()
torch.cuda.synchronize()
out = torch.empty(n, dtype=torch.uint8, device="cuda")
for batch_mib in [8192, 4096, 1024, 512, 256, 128, 96, 64, 32]:
    bsz = batch_mib << 20
    nb = n // bsz
    K = bsz // bb
    blks = None
    def run():
        global blks
        if blks is None:
            blks = [svc.compress_device(t[i * bsz:(i + 1) * bsz], bb) for i in range(min(nb, 4))]
        for i in range(nb):
            svc.compress_device(t[i * bsz:(i + 1) * bsz], bb, out=blks[i % len(blks)])
    run(); torch.cuda.synchronize()
    svc.ctx.reset_profiling(); svc.ctx.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(3): run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    svc.ctx.set_profiling(False)
    ks = {name: svc.ctx.kernel_time(kid)[0] / 3 for kid, name in pkg.native.KERNEL_NAMES.items()}
    print("%s batch %5d MiB (%4d calls): %.2f ms  2N/t = %.0f GB/s (%.2f of 8 TB/s) | k1 %.2f k2 %.2f k3 %.2f ms" % (
        kind, batch_mib, nb, dt * 1e3, 2 * n / dt / 1e9, 2 * n / dt / 8e12, ks["k1_histogram"], ks["k2_codebuild"], ks["k3_encode"]), flush=True)
    del blks
