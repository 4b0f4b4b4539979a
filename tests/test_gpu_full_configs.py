"""BASELINE.json's configurations at their FULL sizes through the C ABI (needs a real MI355X), every output hashed and every status
word compared with the oracle's: config 3 as written (ONE stream of 1 GiB, window 22, many metablocks) by a gang of blocks and by one
block; the metric's batch (256 x 4 MiB) and SURVEY 8(a1)'s make-up (a quarter seed); and -- where the box has two devices -- the
sharded decode of rust-brotli-decompressor_amd/sharding.py over two `nccl` ranks with the real device decode.

What the reference does for each stream is src/decode.rs:2330-2744 (ProcessCommandsInternal) under BrotliDecompressStream."""
import hashlib
import json
import os
import socket
import sys

import pytest

import oracle_lib as oracle
from conftest import ROOT

pytestmark = pytest.mark.gpu

sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def _w():
    import workloads as w
    if not w.encoder_available():
        pytest.fail("libbrotlienc is not available: the GPU suite needs the encoder of the image for its synthetic streams")
    return w


def _with_env(name, value, fn):
    old = os.environ.get(name)
    try:
        if value is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = value
        return fn()
    finally:
        if old is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = old


def test_one_stream_of_1_gib_by_a_gang_and_by_one_block(pkg):
    """BASELINE config 3 as written.  The oracle's status words (result, code, size, consumed bytes, commands, metablocks) and SHA-256
    of the regenerated data, through a gang of sixteen blocks, of eight (BROTLI_AMD_GANG=8) and by one block (BROTLI_AMD_GANG=0)."""
    w = _w()
    raw = w.long_backref_stream(1000, 1 << 30)
    want = hashlib.sha256(raw).digest()
    c = w.brotli_compress(raw, 5, 22)
    del raw
    info, out = oracle.decode(c, 1 << 30, 1)
    assert info.result == 1 and info.decoded_size == 1 << 30 and hashlib.sha256(out).digest() == want and info.num_metablocks > 100
    del out

    def run():
        b = pkg.Batch(1)
        res, outs = b.decode_host([c], [1 << 30], 1)
        gang = b.last_gang()
        b.close()
        r = res[0]
        got = (r.result, r.error_code, r.decoded_size, r.consumed, r.num_commands, r.num_metablocks, hashlib.sha256(outs[0]).digest())
        return got, gang, r.engine_commands

    exp = (info.result, info.error_code, info.decoded_size, info.consumed, info.num_commands, info.num_metablocks, want)
    for env, blocks in ((None, 16), ("8", 8), ("0", 1)):   # (a long stream on a device with the CUs to spare: sixteen blocks; eight; one)
        got, gang, eng = _with_env("BROTLI_AMD_GANG", env, run)
        assert gang == blocks and got == exp and eng >= 0.95 * info.num_commands, (env, gang, got[:6], exp[:6], eng)


@pytest.mark.parametrize("kind,seed0", [("long_backref", 1000), ("survey_mix", 4000)])
def test_the_metric_batch_every_output_hashed(pkg, kind, seed0):
    """256 x 4 MiB -- the batch the metric is quoted on (bench.py's default workload, the same seeds) and the survey's make-up
    beside it: 256 distinct streams, every output hashed, every status word the oracle's."""
    w = _w()
    us = w.make_streams(kind, 256, 4 << 20, seed0)
    datas, caps = [c for c, _, _ in us], [sz for _, sz, _ in us]
    b = pkg.Batch(len(datas))
    res, outs = b.decode_host(datas, caps, 1)
    assert b.last_gang() == 1
    b.close()
    bad = []
    for i, ((c, sz, sha), r, o) in enumerate(zip(us, res, outs)):
        info, _ = oracle.decode(c, sz, 1)
        if ((r.result, r.error_code, r.decoded_size, r.consumed, r.num_commands, r.num_metablocks) !=
                (info.result, info.error_code, info.decoded_size, info.consumed, info.num_commands, info.num_metablocks)
                or hashlib.sha256(o).hexdigest() != sha or info.result != 1):
            bad.append(i)
    assert not bad, bad[:8]
    assert sum(r.engine_commands for r in res) >= 0.95 * sum(r.num_commands for r in res)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _nccl_worker(rank, world, port, q):
    """one rank of the sharded decode: the real device decode on cuda:rank, the collectives over RCCL"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    from conftest import load_pkg
    import importlib.util
    spec = importlib.util.spec_from_file_location("sharding", os.path.join(ROOT, "rust-brotli-decompressor_amd", "sharding.py"))
    sharding = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharding)
    pkg = load_pkg()
    names, streams, caps = _sharded_batch()

    def decode_fn(ss, cc):
        if not ss:
            return np.zeros((0, 4), dtype=np.int64), []
        b = pkg.Batch(len(ss))
        res, outs = b.decode_host(ss, cc, 1)
        b.close()
        return np.array([[r.result, r.error_code, r.decoded_size, r.consumed] for r in res], dtype=np.int64).reshape(-1, 4), outs

    mine, outs, status = sharding.decode_sharded(streams, caps, decode_fn, device=torch.device("cuda", rank))
    q.put((rank, mine, [hashlib.sha256(o).hexdigest() for o in outs], status.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def _sharded_batch():
    names = ["alice29.txt.compressed", "zeros.compressed", "borked.compressed", "monkey.compressed", "lcet10.txt.compressed",
             "empty.compressed", "random_org_10k.bin.compressed", "metablock_reset.compressed", "x.compressed", "plrabn12.txt.compressed"]
    manifest = {e["name"]: e for e in json.load(open(os.path.join(GOLD, "manifest.json")))}
    streams = [open(os.path.join(GOLD, "testdata", n), "rb").read() for n in names]
    caps = [manifest[n].get("size", 64) + 16 for n in names]
    import workloads as w
    if w.encoder_available():   # (and some of the metric's streams, so that the parts are not all tiny)
        for c, sz, _ in w.make_streams("long_backref", 6, 1 << 20, 3000):
            names.append("long_backref"); streams.append(c); caps.append(sz)
    return names, streams, caps


def test_two_nccl_ranks_shard_a_batch_with_the_device_decode():
    """SURVEY 8(e) with two real devices: broadcast of the descriptor table, LPT partition, every rank decodes its part on its own
    GPU, the status words gathered (RCCL).  Skipped on a box with one device (the driver's 8-GPU node runs it)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two devices")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    names, streams, caps = _sharded_batch()
    (_, mine0, sha0, status0), (_, mine1, sha1, status1) = got
    assert sorted(mine0 + mine1) == list(range(len(names))) and not set(mine0) & set(mine1)
    assert status0 == status1
    shas = dict(zip(mine0, sha0)); shas.update(zip(mine1, sha1))
    for i, (s, cap) in enumerate(zip(streams, caps)):
        info, exp = oracle.decode(s, cap, 1)
        assert status0[i] == [info.result, info.error_code, info.decoded_size, info.consumed], (names[i], status0[i])
        assert shas[i] == hashlib.sha256(exp).hexdigest(), names[i]
