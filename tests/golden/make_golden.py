"""Writes tests/golden/reference_vectors.json: the known-answer vectors the reference itself holds for the
hot path, as DATA (inputs as recipes or hex, expected outputs, and where the reference pins each one).

There is no JVM in the build image, so none of these come from running the reference; each is either
asserted by one of the reference's own unit tests or was printed by the reference into its own logs
(paths relative to /root/reference).  test_small.bin / test_input.bin next to this file are the
reference's root-level data files (test_2mb.bin is 2 MiB of 0x41 and is regenerated from its SHA-256).
Run:  python tests/golden/make_golden.py
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))

VECTORS = {
    "histogram_kats": [  # app/src/test/java/com/datacomp/service/cpu/CpuFrequencyServiceTest.java
        {"src": "CpuFrequencyServiceTest.java:25-35", "data": [0, 1, 2, 1, 0, 1], "offset": 0, "length": 6,
         "expect": {"0": 2, "1": 3, "2": 1}},
        {"src": "CpuFrequencyServiceTest.java:38-49", "recipe": "identity256", "offset": 0, "length": 256,
         "expect_all": 1},
        {"src": "CpuFrequencyServiceTest.java:70-80", "recipe": "fifty5_fifty10", "offset": 25, "length": 50,
         "expect": {"5": 25, "10": 25}},
        {"src": "CpuFrequencyServiceTest.java:83-91", "data": [255, 254, 253, 255], "offset": 0, "length": 4,
         "expect": {"255": 2, "254": 1, "253": 1}},
    ],
    "concat_kats": [  # app/src/test/java/com/datacomp/service/gpu/ReductionBasedEncodingTest.java
        {"src": "ReductionBasedEncodingTest.java:122-163", "lengths": {"65": 1, "66": 2, "67": 3, "68": 3},
         "codes": {"65": 0, "66": 2, "67": 6, "68": 7}, "text": "ABAACDAB", "bits": 14, "payload_hex": "46e8"},
        {"src": "ReductionBasedEncodingTest.java:27-41", "merge": [[5, 3], [3, 3]], "value": 43},
    ],
    "payload_pins": [  # sizes the reference logged for its own test inputs: app/logs/datacomp-2025-11-14.log
        {"name": "abcd16", "recipe": "AAAABBBBCCCCDDDD", "chunk_mb": 16, "file_name": "small_input.bin",
         "payload_size": 4, "payload_hex": "0055aaff", "file_size": 667, "src": "log:297,312; SURVEY appendix C"},
        {"name": "a1024", "recipe": "A*1024", "payload_size": 128, "payload_all_zero": True,
         "src": "log:371,386"},
        {"name": "hello", "recipe": "Hello World! *100", "payload_size": 500, "file_name": "test.txt",
         "file_size": 1156, "src": "log:134,146"},
        {"name": "integrity", "recipe": "Test data for integrity check", "payload_size": 15, "file_name": "test.txt",
         "file_size": 671, "src": "log:203"},
        {"name": "random1024", "recipe": "java.util.Random(42).nextBytes(1024)", "payload_size": 1008,
         "src": "log:334,349"},
        {"name": "random10240", "recipe": "java.util.Random(42).nextBytes(10240)", "payload_size": 10239,
         "file_name": "random.bin", "file_size": 10897, "src": "log:109,121"},
        {"name": "mod256_3mib", "recipe": "i%256 for 3 MiB, 1 MiB chunks", "payload_size": 3145728,
         "payload_equals_input": True, "src": "log:176,188"},
        {"name": "speed512k", "recipe": "'A'+(i/100)%26 for 512 KiB", "file_name": "speed_test_input.bin",
         "payload_size": 312530, "file_size": 313198, "src": "log:224; Phase3IntegrationTest.java:99-142"},
        {"name": "a2mib", "recipe": "A*2097152 (test_2mb.bin)", "payload_size": 262144, "payload_all_zero": True,
         "sha256": "5b766f6d76a999636fd93b4e039d5a32187f84a19c0950449f0c721da0223914", "src": "log:259,274"},
        {"name": "test_small", "file": "test_small.bin", "payload_size": 256, "payload_all_zero": True,
         "sha256": "3a34c8dc4aec1554c04e0d0e61179d08362b329029db4632f5f086c37be74caa", "src": "SURVEY section 4"},
        {"name": "test_input", "file": "test_input.bin", "payload_size": 1048576, "payload_equals_input": True,
         "sha256": "fcd8ad1070c6f303848f387385e8f1204be2e2b1a832ea8ce7c12d3ec983c37d", "src": "SURVEY section 4"},
    ],
    "container": {"magic": "44435a46", "version": 1, "chunk_meta_bytes": 572, "fixed_header_bytes": 68,
                  "empty_file": {"file_name": "empty.txt", "file_size": 85},
                  "src": "CompressionHeader.java:51-85; log:159 (0 -> 85 bytes = 68 + 9 + 8)"},
}

if __name__ == "__main__":
    with open(os.path.join(HERE, "reference_vectors.json"), "w") as f:
        json.dump(VECTORS, f, indent=1, sort_keys=True)
    print("wrote reference_vectors.json")
