// dcz_cli.cpp -- command line with the surface of cli/DataCompCLI.java:24-146:
//   dczcli compress|c|decompress|d <input> <output> [chunkMB] [--gpus N]   (default chunk 32 MB, DataCompCLI.java:35)
// plus `verify <file.dcz>`, `histogram <file>`, `bench <file> [chunkMB] [--gpus N] [--cpu-mbps X]`
// (benchmark/BenchmarkSuite.java:37-121: 3 warm-ups, 5 timed runs, MB/s = bytes / 1e6 / s, for compress AND decompress,
// the StageMetrics summary, a byte-for-byte check of the round trip, and BenchmarkComparison's summary: the CPU line and
// "GPU Speedup" need a CPU figure, which this product cannot produce -- it holds no CPU codec -- so it is handed in:
// --cpu-mbps = MB/s of the reference's CpuCompressionService on the same file, or of bench.py's cpu_baseline) and
// `shardplan <chunks> <gpus>` (the chunk ranges --gpus N gives each device; needs no GPU).
// --gpus N shards the file's chunks over N devices of the node (contiguous ranges, one pipeline per device).  The reference CLI is hard-wired to the CPU service
// (DataCompCLI.java:62); this one runs the HIP service and fails loudly when no gfx950 device is present.
#include <sys/stat.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "dcz_service.h"

static void usage() {
    std::fprintf(stderr,
                 "Usage: dczcli <operation> <input> <output> [chunkMB] [--gpus N]\n"
                 "  operations: compress | c | decompress | d | verify <file> | histogram <file>\n"
                 "              bench <file> [chunkMB] [--gpus N] [--cpu-mbps X] | shardplan <chunks> <gpus>\n");
}

static std::string fmt_size(long long b) {
    char buf[64];
    if (b < 1024) std::snprintf(buf, sizeof buf, "%lld B", b);
    else if (b < 1024 * 1024) std::snprintf(buf, sizeof buf, "%.2f KB", b / 1024.0);
    else if (b < 1024LL * 1024 * 1024) std::snprintf(buf, sizeof buf, "%.2f MB", b / (1024.0 * 1024));
    else std::snprintf(buf, sizeof buf, "%.2f GB", b / (1024.0 * 1024 * 1024));
    return buf;
}

static long long file_size(const std::string& p) {
    struct stat st;
    return ::stat(p.c_str(), &st) == 0 ? (long long)st.st_size : -1;
}

// benchmark/BenchmarkSuite.java:68-121 (benchmarkService) for this service, plus the decompress timing upstream lacks
static bool same_bytes(const std::string& a, const std::string& b) {
    std::ifstream fa(a, std::ios::binary), fb(b, std::ios::binary);
    if (!fa || !fb) return false;
    std::vector<char> ba(1 << 20), bb(1 << 20);
    while (true) {
        fa.read(ba.data(), (std::streamsize)ba.size());
        fb.read(bb.data(), (std::streamsize)bb.size());
        if (fa.gcount() != fb.gcount()) return false;
        if (fa.gcount() == 0) return true;
        if (std::memcmp(ba.data(), bb.data(), (size_t)fa.gcount()) != 0) return false;
    }
}

static int run_bench(const std::string& in, int chunkMB, int gpus, double cpuMBps) {
    const long long inSize = file_size(in);
    if (inSize < 0) {
        std::fprintf(stderr, "Error: Input file does not exist: %s\n", in.c_str());
        return 1;
    }
    const int warmup = 3, iters = 5;  // application.conf:50,53
    const std::string dcz = in + ".bench.dcz", back = in + ".bench.out";
    datacomp::HipCompressionService svc(chunkMB);
    if (!svc.isAvailable()) throw datacomp::IOError("no gfx950 device available");
    datacomp::StageMetrics cm, dm;
    auto comp = [&]() {
        if (gpus > 0) datacomp::compressSharded(in, dcz, chunkMB, gpus, {}, &cm);
        else { svc.compress(in, dcz); cm = svc.getLastStageMetrics(); }
    };
    auto decomp = [&]() {
        if (gpus > 0) datacomp::decompressSharded(dcz, back, gpus, {}, &dm);
        else { svc.decompress(dcz, back); dm = svc.getLastStageMetrics(); }
    };
    std::printf("Benchmarking %s: %s, %d MB chunks%s\n", in.c_str(), svc.getServiceName().c_str(), chunkMB,
                gpus > 0 ? (", " + std::to_string(gpus) + " GPU(s)").c_str() : "");
    for (int i = 0; i < warmup; i++) comp();
    double ct = 0, dt = 0;
    for (int i = 0; i < iters; i++) {
        const auto t0 = std::chrono::steady_clock::now();
        comp();
        ct += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    const long long outSize = file_size(dcz);
    for (int i = 0; i < warmup; i++) decomp();
    for (int i = 0; i < iters; i++) {
        const auto t0 = std::chrono::steady_clock::now();
        decomp();
        dt += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    const bool same = file_size(back) == inSize && same_bytes(in, back);
    ct /= iters;
    dt /= iters;
    std::printf("Benchmark complete:\n  input %lld bytes, output %lld bytes (ratio %.2f%%), %d iterations after %d warm-ups\n"
                "  compress:   %.3f s avg, %.2f MB/s\n  decompress: %.3f s avg, %.2f MB/s%s\n",
                inSize, outSize, inSize ? 100.0 * outSize / inSize : 0.0, iters, warmup, ct, ct > 0 ? inSize / 1e6 / ct : 0.0, dt,
                dt > 0 ? inSize / 1e6 / dt : 0.0, same ? "  (round trip byte-identical)" : "  (ROUND TRIP DIFFERS FROM THE INPUT)");
    std::printf("\n[compress] %s\n[decompress] %s", cm.summary().c_str(), dm.summary().c_str());
    // BenchmarkSuite.BenchmarkComparison.getSummary (benchmark/BenchmarkSuite.java:152-170), compress throughput
    const double mbps = ct > 0 ? inSize / 1e6 / ct : 0.0;
    std::printf("\n=== Benchmark Results ===\n");
    if (cpuMBps > 0) std::printf("CPU (figure handed in with --cpu-mbps): %.2f MB/s (%.3fs)\n", cpuMBps, inSize / 1e6 / cpuMBps);
    std::printf("%s: %.2f MB/s (%.3fs)\n", svc.getServiceName().c_str(), mbps, ct);
    if (cpuMBps > 0) std::printf("GPU Speedup: %.2fx\n", mbps / cpuMBps);
    else std::printf("CPU leg: none in this product (no CPU codec); pass --cpu-mbps, or see bench.py's cpu_baseline\n");
    std::remove(dcz.c_str());
    std::remove(back.c_str());
    return same ? 0 : 2;
}

int main(int argc, char** argv) {
    // --gpus N anywhere on the line
    int gpus = 0;
    double cpuMBps = 0;
    std::vector<char*> av;
    for (int i = 0; i < argc; i++) {
        if (std::strcmp(argv[i], "--cpu-mbps") == 0 && i + 1 < argc) {
            cpuMBps = std::atof(argv[++i]);
            continue;
        }
        if (std::strcmp(argv[i], "--gpus") == 0 && i + 1 < argc) {
            gpus = std::atoi(argv[++i]);
            if (gpus < 1) {
                std::fprintf(stderr, "Invalid --gpus value\n");
                return 1;
            }
        } else {
            av.push_back(argv[i]);
        }
    }
    argc = (int)av.size();
    argv = av.data();
    if (argc < 3) {
        usage();
        return 1;
    }
    const std::string op = argv[1], in = argv[2];
    try {
        if (op == "shardplan") {  // pure host logic
            if (argc < 4) {
                usage();
                return 1;
            }
            for (auto& pr : datacomp::planShards(std::atoll(argv[2]), std::atoi(argv[3])))
                std::printf("%lld %lld\n", (long long)pr.first, (long long)pr.second);
            return 0;
        }
        if (op == "bench") {
            int chunkMB = 32;
            if (argc > 3) chunkMB = std::atoi(argv[3]);
            if (chunkMB <= 0) {
                std::fprintf(stderr, "Invalid chunk size: %s\n", argv[3]);
                return 1;
            }
            return run_bench(in, chunkMB, gpus, cpuMBps);
        }
        if (op == "verify") {
            datacomp::HipCompressionService svc(32);
            if (!svc.isAvailable()) throw datacomp::IOError("no gfx950 device available");
            const bool ok = svc.verifyIntegrity(in);
            std::printf("%s\n", ok ? "OK" : "CORRUPT");
            return ok ? 0 : 2;
        }
        if (op == "histogram") {
            datacomp::HipFrequencyService fs;
            if (!fs.isAvailable()) throw datacomp::IOError("no gfx950 device available");
            std::ifstream f(in, std::ios::binary | std::ios::ate);
            if (!f) throw datacomp::IOError("Input file does not exist: " + in);
            std::vector<uint8_t> d((size_t)f.tellg());
            f.seekg(0);
            f.read(reinterpret_cast<char*>(d.data()), (std::streamsize)d.size());
            const auto h = fs.computeHistogram(d.data(), 0, d.size());
            for (int i = 0; i < 256; i++) std::printf("%d %lld\n", i, (long long)h[(size_t)i]);
            return 0;
        }
        if (argc < 4) {
            usage();
            return 1;
        }
        const std::string out = argv[3];
        int chunkMB = 32;
        if (argc > 4) {
            char* end = nullptr;
            chunkMB = (int)std::strtol(argv[4], &end, 10);
            if (!end || *end) {
                std::fprintf(stderr, "Invalid chunk size: %s\n", argv[4]);
                return 1;
            }
        }
        if (file_size(in) < 0) {
            std::fprintf(stderr, "Error: Input file does not exist: %s\n", in.c_str());
            return 1;
        }
        datacomp::HipCompressionService svc(chunkMB);
        if (!svc.isAvailable()) throw datacomp::IOError("no gfx950 device available");
        const auto t0 = std::chrono::steady_clock::now();
        auto progress = [](double p) { std::printf("\rProgress: %d%%", (int)(p * 100)); std::fflush(stdout); };
        datacomp::StageMetrics sharded;
        if (op == "compress" || op == "c") {
            std::printf("Compressing...\n  Input:  %s\n  Output: %s\n  Size:   %s\n", in.c_str(), out.c_str(),
                        fmt_size(file_size(in)).c_str());
            if (gpus > 0) datacomp::compressSharded(in, out, chunkMB, gpus, progress, &sharded);
            else svc.compress(in, out, progress);
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const long long a = file_size(in), b = file_size(out);
            std::printf("\n\nCompression complete!\n  Original size:   %s\n  Compressed size: %s\n  Compression ratio: %.2f%%\n"
                        "  Time: %.2f seconds\n  Throughput: %.2f MB/s\n",
                        fmt_size(a).c_str(), fmt_size(b).c_str(), a ? 100.0 * b / a : 0.0, sec, sec > 0 ? a / 1e6 / sec : 0.0);
        } else if (op == "decompress" || op == "d") {
            std::printf("Decompressing...\n  Input:  %s\n  Output: %s\n", in.c_str(), out.c_str());
            if (gpus > 0) datacomp::decompressSharded(in, out, gpus, progress, &sharded);
            else svc.decompress(in, out, progress);
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const long long a = file_size(in), b = file_size(out);
            std::printf("\n\nDecompression complete!\n  Compressed size:   %s\n  Decompressed size: %s\n  Time: %.2f seconds\n"
                        "  Throughput: %.2f MB/s\n",
                        fmt_size(a).c_str(), fmt_size(b).c_str(), sec, sec > 0 ? b / 1e6 / sec : 0.0);
        } else {
            std::fprintf(stderr, "Unknown operation: %s\n", op.c_str());
            usage();
            return 1;
        }
        std::printf("\n%s", (gpus > 0 ? sharded : svc.getLastStageMetrics()).summary().c_str());
        return 0;
    } catch (const datacomp::IOError& e) {
        std::fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Unexpected error: %s\n", e.what());
        return 1;
    }
}
