"""Which streams have metablocks whose prefix-code tables do not fit the LDS arena (slower instantiation of the loop)?
usage: python tools/spill_report.py [lds_arena_bytes ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from conftest import load_pkg
import param_corpus
pkg = load_pkg()
G = os.path.join(ROOT, "tests", "golden")
m = [e for e in json.load(open(os.path.join(G, "manifest.json"))) if e["name"] != "rnd_chunk.br" and not e.get("must_fail")]
names = [e["name"] for e in m]
datas = [open(os.path.join(G, "testdata", n), "rb").read() for n in names]
caps = [e["size"] + 16 for e in m]
for label, comp, raw in param_corpus.corpus():
    names.append(label); datas.append(comp); caps.append(len(raw) + 16)
for arena in [int(a) for a in sys.argv[1:]] or [0]:
    b = pkg.Batch(len(datas), lds_arena_bytes=arena)
    res, _ = b.decode_host(datas, caps, pkg.FLAG_LARGE_WINDOW)
    b.close()
    sp = [(n, r.spilled_metablocks, r.num_metablocks) for n, r in zip(names, res) if r.spilled_metablocks]
    print("arena %d: %d of %d streams spill:" % (arena, len(sp), len(names)), " ".join("%s(%d/%d)" % x for x in sp))
