#!/usr/bin/env python3
"""Writes the committed edition of the parameterised encoder corpus (tests/param_corpus.py at one fifth of its data sizes):
tests/golden/param_corpus/<n>.br + manifest.json (label, file, compressed size, raw size, SHA-256 of the raw data).
Needs libbrotlienc (this build container has it); every stream is checked with libbrotlidec before it is written."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libbrotli_ref as ref  # noqa: E402
import param_corpus  # noqa: E402

out_dir = param_corpus.COMMITTED
os.makedirs(out_dir, exist_ok=True)
for f in os.listdir(out_dir):
    os.remove(os.path.join(out_dir, f))
manifest = []
for i, (label, comp, raw) in enumerate(param_corpus.corpus(scale=0.2)):
    r = ref.decode(comp, len(raw) + 16, True)
    assert r[0] == 1 and r[2] == raw, label
    name = "%03d.br" % i
    open(os.path.join(out_dir, name), "wb").write(comp)
    manifest.append({"label": label, "file": name, "csize": len(comp), "size": len(raw), "sha256": hashlib.sha256(raw).hexdigest()})
json.dump(manifest, open(os.path.join(out_dir, "manifest.json"), "w"), indent=0)
print(len(manifest), "streams,", sum(e["csize"] for e in manifest), "bytes")
