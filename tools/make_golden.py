#!/usr/bin/env python3
"""Build tests/golden/ from the reference's own test DATA (run in the build container only).

Everything written here is data -- inputs and expected outputs -- never source text:
  tests/golden/testdata/<name>          the reference's compressed fixtures (testdata/*.compressed*, *.br, *.bro)
  tests/golden/manifest.json            name, compressed size, expected size + SHA-256/CRC-32 of the paired original
  tests/golden/inline_vectors.json      byte-array vectors that the reference's tests hold inline, with the
                                        result / output / error code those tests assert (file:line cited per vector)
  tests/golden/huffman_tables.json      known-answer lookup tables of src/huffman/tests.rs, re-expressed as
                                        (code lengths -> every table entry)
  tests/golden/rnd_chunk_edges.json     prefix/postfix bytes + size for the large-window fixture
The GPU box never sees /root/reference; tests read only tests/golden/.
"""
import hashlib, json, os, re, shutil, sys, zlib

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def lines(path, a, b):
    with open(os.path.join(REF, path), newline="\n") as f:
        ls = f.readlines()
    return "".join(ls[a - 1:b])


def byte_array(text):
    """first `= [ints...];` / `= {ints...};` initialiser in text (at least two integer literals or hex)"""
    for m in re.finditer(r"=\s*&?\s*[\[{]([^\[\]{}]*?)[\]}]\s*;", text, re.S):
        body = m.group(1)
        if ";" in body:
            continue
        toks = re.findall(r"(0[xX][0-9a-fA-F]+|\d+)(?:u8)?", body)
        if toks:
            return bytes(int(t, 0) for t in toks)
    raise ValueError("no byte array in: " + text[:200])


def fn_body(path, name, c_style=False):
    text = open(os.path.join(REF, path), newline="\n").read()
    m = re.search((r"\bvoid %s\(" if c_style else r"\bfn %s\(") % name, text)
    assert m, (path, name)
    end = text.index("\n}\n", m.start())
    return text[m.start():end]


def testdata():
    src = os.path.join(REF, "testdata")
    dst = os.path.join(GOLD, "testdata")
    os.makedirs(dst, exist_ok=True)
    manifest = []
    for name in sorted(os.listdir(src)):
        orig = None
        if ".compressed" in name:
            orig = name[:name.index(".compressed")]
        elif name.endswith(".bro"):
            orig = name[:-4] + ".unbro"
        elif name == "random1024.br":
            orig = "random1024"
        elif name == "rnd_chunk.br":
            orig = None
        else:
            continue
        shutil.copyfile(os.path.join(src, name), os.path.join(dst, name))
        entry = {"name": name, "csize": os.path.getsize(os.path.join(src, name))}
        if name == "borked.compressed":
            entry["must_fail"] = True  # src/bin/integration_tests.rs:971-978
        elif name == "rnd_chunk.br":
            entry["large_window"] = True
            entry["size"] = 100011280  # src/bin/integration_tests.rs:997-1006
        else:
            data = open(os.path.join(src, orig), "rb").read()
            entry.update(size=len(data), sha256=hashlib.sha256(data).hexdigest(), crc32=zlib.crc32(data), original=orig)
        manifest.append(entry)
    json.dump(manifest, open(os.path.join(GOLD, "manifest.json"), "w"), indent=1)
    pre = open(os.path.join(src, "rnd_prefix"), "rb").read()
    post = open(os.path.join(src, "rnd_postfix"), "rb").read()
    json.dump({"size": 100011280, "zero_count": 100000000, "prefix_hex": pre.hex(), "postfix_hex": post.hex(),
               "cite": "src/bin/integration_tests.rs:997-1006"},
              open(os.path.join(GOLD, "rnd_chunk_edges.json"), "w"))
    return manifest


FOX = b"The quick brown fox jumps over the lazy dog"


def inline_vectors():
    v = []

    def add(name, cite, data, **exp):
        v.append(dict(name=name, cite=cite, input_hex=data.hex(), **exp))

    c_ok = byte_array(fn_body("c/main.c", "simple_test", True))
    add("c_main_simple", "c/main.c:17-33", c_ok, result=1, output_hex=(b"THIS IS A TEST OF THE EMERGENCY BROADCAST SYSTEM\n").hex())
    c_bad = byte_array(fn_body("c/main.c", "negative_test", True))
    add("c_main_negative", "c/main.c:57-81", c_bad, result=0, error_code=-8)
    add("test_10x10y", "src/test.rs:176-196", byte_array(fn_body("src/test.rs", "test_10x10y")), result=1,
        output_hex=(b"X" * 10 + b"Y" * 10).hex(), consumed_all=True)
    add("test_x", "src/test.rs:199-211", byte_array(fn_body("src/test.rs", "test_x")), result=1, output_hex=b"X".hex(), consumed_all=True)
    add("test_corrupt_input_large_distance_code", "src/test.rs:214-223", byte_array(fn_body("src/test.rs", "test_corrupt_input_large_distance_code")), result=0,
        large_window=True)
    add("test_empty", "src/test.rs:226-238", byte_array(fn_body("src/test.rs", "test_empty")), result=1, output_hex="", consumed_all=True)
    qf = byte_array(fn_body("src/test.rs", "test_quickfox_repeated_custom"))
    add("test_quickfox_repeated", "src/test.rs:244-272", qf, result=1, output_sha256=hashlib.sha256(FOX * 4096).hexdigest(),
        output_size=176128, consumed_all=True)
    add("test_early_eof", "src/test.rs:410-421", byte_array(fn_body("src/test.rs", "test_early_eof")), result=0, large_window=True)
    add("test_run_out_of_writer_space_valid", "src/test.rs:424-435", byte_array(fn_body("src/test.rs", "test_run_out_of_writer_space")), result=1,
        output_hex=(b"\0" * 2048).hex())
    for k, (cite, out) in enumerate([("903-912", b"himselfself"), ("916-926", b"scrollroll"), ("929-938", b"leftdatadataleft")]):
        body = fn_body("src/bin/integration_tests.rs", "test_intact_distance_ring_buffer%d" % k)
        assert ('b"%s"' % out.decode()) in body
        add("test_intact_distance_ring_buffer%d" % k, "src/bin/integration_tests.rs:" + cite, byte_array(body), result=1, output_hex=out.hex())
    enc = b"\x1b\x03)\x00\xa4\xcc\xde\xe2\xb3 vA\x00\x0c"
    assert 'b"\\x1b\\x03)\\x00\\xa4\\xcc\\xde\\xe2\\xb3 vA\\x00\\x0c"' in lines("src/bin/error_handling_tests.rs", 1, 12)
    add("error_handling_encoded", "src/bin/error_handling_tests.rs:7", enc, result=1)
    valid = byte_array(fn_body("src/bin/error_handling_tests.rs", "test_padding_2_rejection"))
    sentence = b"the quick brown fox jumps over the lazy dog twice for redundancy and length"
    add("padding2_valid", "src/bin/error_handling_tests.rs:184-198", valid, result=1, output_hex=sentence.hex())
    for off, x in [(13, 0x01), (23, 0x01), (33, 0x55)]:
        c = bytearray(valid); c[off] ^= x
        add("padding2_flip_%d_%02x" % (off, x), "src/bin/error_handling_tests.rs:204-221", bytes(c), result=0, error_code=-15)
    hx = "".join(re.findall(r'\b([0-9a-f]{8,})', fn_body("src/bin/error_handling_tests.rs", "test_rejects_metablock_length_overflow")))
    add("metablock_length_overflow", "src/bin/error_handling_tests.rs:230-265", bytes.fromhex(hx), result=0, error_code=-10)
    # all 256 one-byte streams: src/bin/tests.rs:76-98 -- exactly these succeed
    ok = {6, 26, 51, 53, 55, 57, 59, 61, 63}
    for b in range(256):
        add("one_byte_%02x" % b, "src/bin/tests.rs:76-98", bytes([b]), result_is_success=(b in ok), reader_eof=True)
    json.dump(v, open(os.path.join(GOLD, "inline_vectors.json"), "w"), indent=0)
    return v


def huffman():
    path = "src/huffman/tests.rs"
    text = open(os.path.join(REF, path)).read()
    fns = re.split(r"#\[test\]\s*fn ", text)[1:]
    out = []
    for fn in fns:
        name = fn[:fn.index("(")]
        entries = re.search(r"let end_table[^=]*=\s*\[(.*?)\];", fn, re.S).group(1)
        pairs = re.findall(r"HuffmanCode\s*\{\s*bits:\s*(\d+),\s*value:\s*(\d+),?\s*\}\s*(;\s*(\d+))?", entries)
        table = []
        for bits, value, _, rep in pairs:
            table += [[int(bits), int(value)]] * (int(rep) if rep else 1)
        rec = {"name": name, "cite": path, "table": table}
        if name.startswith("code_length"):
            rec["kind"] = "code_lengths"
            rec["code_lengths"] = list(byte_array(re.search(r"let code_lengths[^=]*(=\s*\[.*?\];)", fn, re.S).group(1)))[:18]
        elif name.startswith("simple"):
            rec["kind"] = "simple"
            rec["symbols"] = [int(x) for x in re.search(r"let mut val[^=]*=\s*\[(.*?)\];", fn, re.S).group(1).replace(" ", "").split(",") if x]
            rec["num_symbols"] = int(re.search(r"BrotliBuildSimpleHuffmanTable\(&mut table, 8, &mut val, (\d+)\)", fn).group(1))
        else:
            rec["kind"] = "full"
            arr = [int(x) for x in re.split(r"[,\s]+", re.search(r"let symbol_array[^=]*=\s*\[(.*?)\];", fn, re.S).group(1).strip()) if x]
            counts = [int(x) for x in re.search(r"let mut counts[^=]*=\s*\[(.*?)\];", fn, re.S).group(1).replace(" ", "").split(",") if x]
            size = int(re.search(r"assert_eq!\(size, (\d+)\)", fn).group(1))
            lengths = [0] * 704
            for bits in range(1, len(counts)):
                idx = bits - 16
                prev = -1
                for _ in range(counts[bits]):
                    idx = arr[16 + idx]
                    assert idx > prev, "symbols must ascend within one length"
                    prev = idx
                    lengths[idx] = bits
            n = max(i for i, l in enumerate(lengths) if l) + 1
            rec.update(code_lengths=lengths[:n], root_bits=8, size=size)
            assert len(table) >= size, (name, len(table), size)
            rec["table"] = table[:size]
        out.append(rec)
    json.dump(out, open(os.path.join(GOLD, "huffman_tables.json"), "w"))
    return out


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; golden files are already committed")
    os.makedirs(GOLD, exist_ok=True)
    m = testdata()
    v = inline_vectors()
    h = huffman()
    print(len(m), "fixtures,", len(v), "inline vectors,", len(h), "huffman tables")
