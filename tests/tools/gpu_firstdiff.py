"""Decode one fixture on the GPU and report the first output byte that differs from the oracle's."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_pkg
import oracle_lib as oracle
pkg = load_pkg()
name = sys.argv[1]
d = open(os.path.join(ROOT, "tests/golden/testdata", name), "rb").read()
info, exp = oracle.decode(d, 1 << 24, 1)
inp = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
out = torch.zeros(len(exp) + 4096, dtype=torch.uint8, device="cuda")
b = pkg.Batch(1)
b.decode_device([inp.data_ptr()], [len(d)], [out.data_ptr()], [len(exp) + 64])
r = b.wait()[0]
got = out.cpu().numpy().tobytes()[:r.produced]
n = min(len(got), len(exp))
i = next((k for k in range(n) if got[k] != exp[k]), None)
print(name, "gpu", r.result, r.error_code, "produced", r.produced, "cmds", r.num_commands, "oracle", info.result, len(exp), "first diff", i)
if i is not None:
    print("exp", exp[max(0, i - 24): i + 24]); print("got", got[max(0, i - 24): i + 24])
