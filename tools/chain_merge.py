#!/usr/bin/env python3
"""How fast does a command parse started at a WRONG bit fall in with a Brotli stream's true chain, and what does the path
engine's first pass have to do per region?  (measure, do not guess.)

Builds tools/chain_merge.c and tools/path_engine_model.c against the oracle (gcc, -DORACLE_STATS) and runs them on streams
of the bench's make-up (workloads.py: C3 long back-references, C4 high-entropy literals) or on files given on the command
line.  CPU only; analysis / test tooling, not product code.

  python tools/chain_merge.py                 # one C3 and one C4 stream of 4 MiB
  python tools/chain_merge.py file.br ...     # these streams
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(tmp):
    dict_path = os.path.join(ROOT, "rust-brotli-decompressor_amd", "data", "dictionary.bin")
    common = ["-O2", "-Wno-unused-function", "-Wno-unused-variable", "-DDICT_PATH=\"%s\"" % dict_path, os.path.join(ROOT, "oracle", "dict_blob.c"), "-lpthread"]
    cm, pm = os.path.join(tmp, "chain_merge"), os.path.join(tmp, "path_engine_model")
    subprocess.check_call(["gcc", "-DORACLE_STATS", "-o", cm, os.path.join(ROOT, "tools", "chain_merge.c"), os.path.join(ROOT, "oracle", "brotli_oracle.c")] + common)
    subprocess.check_call(["gcc", "-o", pm, os.path.join(ROOT, "tools", "path_engine_model.c")] + common)
    return cm, pm


def main():
    with tempfile.TemporaryDirectory() as tmp:
        cm, pm = build(tmp)
        files = sys.argv[1:]
        if not files:
            import workloads as w
            if not w.encoder_available():
                raise SystemExit("no libbrotlienc here: give .br files on the command line")
            for kind, seed in (("long_backref", 1000), ("high_entropy", 2000)):
                c, n, _ = w.make_streams(kind, 1, 4 << 20, seed)[0]
                path = os.path.join(tmp, kind + ".br")
                open(path, "wb").write(c)
                files.append(path)
        for f in files:
            print("=== %s" % os.path.basename(f))
            sys.stdout.flush()
            subprocess.call([cm, f])
            subprocess.call([pm, f])


if __name__ == "__main__":
    main()
