import json,sys
for f in sys.argv[1:]:
    try:
        d=json.load(open(f))
        k=d["kernels"]
        print("%-28s %8.1f GB/s %8.2f ms | k1 %.2f k2 %.2f k3 %.2f k4 %.2f | C/N %.3f" % (f.split("/")[-1], d["value"], d["ms_per_step"], k["k1_histogram"]["avg_ms"], k["k2_codebuild"]["avg_ms"], k["k3_encode"]["avg_ms"], k["k4_decode"]["avg_ms"], d["config"]["compressed_bytes_per_gpu"]/d["config"]["bytes_per_gpu"]))
    except Exception as e:
        print(f, "ERR", e)
