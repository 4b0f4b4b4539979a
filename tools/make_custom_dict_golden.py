#!/usr/bin/env python3
"""tests/golden/custom_dict_vectors.json: the reference's custom-dictionary known-answer tests (src/test.rs:438-520,
test_dict and test_dict_medium) as data -- compressed bytes, dictionary bytes, expected output.  Build container only
(reads /root/reference); the oracle's custom-dictionary entry point is pinned by them (tests/test_oracle.py)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open("/root/reference/src/test.rs").read()


def arr(name, after):
    i = src.index(after)
    m = re.search(r"let\s+(?:mut\s+)?%s\s*:[^=]*=\s*&?\[(.*?)\];" % name, src[i:], re.S)
    return bytes(int(x, 0) for x in re.findall(r"0x[0-9a-fA-F]+|\d+", m.group(1)))


vectors = [
    {"name": "test_dict", "source": "src/test.rs:438-479", "compressed": arr("patch", "fn test_dict()").hex(),
     "dictionary": arr("dict", "fn test_dict()").hex(), "expected": arr("expected", "fn test_dict()").hex()},
    {"name": "test_dict_medium", "source": "src/test.rs:482-520", "compressed": arr("br", "fn test_dict_medium()").hex(),
     "dictionary": bytes(range(256)).hex(), "expected": arr("expected", "fn test_dict_medium()").hex()},
]
json.dump(vectors, open(os.path.join(ROOT, "tests", "golden", "custom_dict_vectors.json"), "w"), indent=1)
print([(v["name"], len(v["compressed"]) // 2, len(v["expected"]) // 2) for v in vectors])
