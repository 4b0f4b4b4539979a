"""Which share of real streams can the command engines take?  python tools/eligibility_survey.py
Encodes a small corpus (the reference's decoded fixtures, the bench's synthetic make-ups, binaries of this image) with the
image's libbrotlienc at several qualities and counts, by uncompressed bytes, the metablocks whose literals do not depend on
context (the engines' condition) and those whose distance contexts use different prefix codes (the scan engine instead of the
path engine).  Builds tools/eligibility_survey.c against the oracle with -DORACLE_STATS."""
import os, subprocess, sys, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import oracle_lib as oracle, libbrotli_ref as ref, workloads as w
exe = os.path.join(ROOT, "tools", "scratch", "eligibility_survey")
os.makedirs(os.path.dirname(exe), exist_ok=True)
subprocess.check_call(["gcc", "-O2", "-Wno-unused-function", "-Wno-unused-variable", "-DORACLE_STATS", "-DDICT_PATH=\"%s\"" % os.path.join(ROOT, "rust-brotli-decompressor_amd", "data", "dictionary.bin"),
                       "-o", exe, os.path.join(ROOT, "tools", "eligibility_survey.c"), os.path.join(ROOT, "oracle", "brotli_oracle.c"), os.path.join(ROOT, "oracle", "dict_blob.c"), "-lpthread"])
corpus = {}
gold = os.path.join(ROOT, "tests", "golden", "testdata")
for n in ("alice29.txt", "asyoulik.txt", "lcet10.txt", "plrabn12.txt"):
    corpus[n] = oracle.decode(open(os.path.join(gold, n + ".compressed"), "rb").read(), 1 << 22, 1)[1]
corpus["long_backref_4MiB"] = w.long_backref_stream(1000, 4 << 20)
for path in ("/usr/bin/python3.10", "/usr/lib/x86_64-linux-gnu/libc.so.6", "/opt/rocm/lib/libamdhip64.so"):
    for p in glob.glob(path + "*")[:1]:
        corpus[os.path.basename(p)] = open(p, "rb").read()[: 8 << 20]
corpus["this_repo_sources"] = b"".join(open(f, "rb").read() for f in sorted(glob.glob(os.path.join(ROOT, "rust-brotli-decompressor_amd", "csrc", "*.h*"))))
import json
corpus["json_like"] = json.dumps([{"id": i, "name": "item%d" % (i * 7919 % 1000), "tags": ["a", "b", "c"][: i % 4], "v": i * 0.37} for i in range(60000)]).encode()
blob = bytearray()
for name, raw in corpus.items():
    for q in (1, 5, 9, 11):
        c = ref.encode(raw, q, 22)
        rec = ("%s_q%d %d\n" % (name, q, len(c))).encode() + c
        blob += rec
out = subprocess.run([exe], input=bytes(blob), capture_output=True)
sys.stdout.write(out.stdout.decode())
