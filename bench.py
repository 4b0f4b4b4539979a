#!/usr/bin/env python3
"""bench.py -- decompressed MB/s of the HIP Brotli decode path on MI355X (BASELINE.json metric).

One "step" = one pass of the decode kernel over one batch of compressed streams that already sit in HBM
(outputs stay in HBM).  Default workload = the configuration the metric is quoted on: a 1 GiB batch of
wbits-22 (4 MiB window) streams, 256 x 4 MiB, synthetic long-back-reference data (SURVEY.md section 8d, C3 as
256 independent streams), 256 distinct streams.

N > 1: one process per GPU (torchrun).  The job is N x 256 streams ("weak" scaling: 1 GiB per GPU).  Rank 0 owns
the descriptor table (one row per stream: compressed size, output capacity, weight), broadcasts it (RCCL), every
rank takes its part by the deterministic longest-processing-time rule of rust-brotli-decompressor_amd/sharding.py
-- the code the 2-rank gloo test covers -- generates the streams of its part locally (they are a function of the
stream index: payload never crosses xGMI), decodes them on its GPU, and the per-stream status words are gathered.

Prints ONE JSON line on rank 0 (contract in the task description), with these extra objects:
  roofline       -- HBM roofline of the decode kernel: algorithmic bytes (compressed read + decompressed written)
                    per launch / mean kernel time from HIP events on the launch stream, against 8 TB/s
  cpu_baseline   -- the CPU oracle ("port": the reference itself is Rust and cannot be built here) on a bounded
                    sample of the same workload: all host threads, one thread, and -- when libbrotlidec.so.1 can
                    be loaded -- Google's C decoder on one thread as a proxy for the reference (a port of it)
  extra_configs  -- (N = 1 only) the other configurations of BASELINE.json, each timed the same way with its own
                    roofline, command rates and CPU legs: C2 1024 x alice29, C4 256 x 4 MiB high-entropy literals, C3 as ONE
                    many-metablock stream (64 MiB, and 1 GiB as BASELINE writes it), the metric's streams twice (512 x 4 MiB),
                    the same data at -q9, as 1024 x 1 MiB, and at the make-up of SURVEY 8(a1)'s prototype
"""
import argparse
import ctypes
import hashlib
import importlib.util
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _load(name, path):
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_pkg():
    return _load("rust_brotli_decompressor_amd", os.path.join(ROOT, "rust-brotli-decompressor_amd", "__init__.py"))


def load_sharding():
    return _load("brotli_amd_sharding", os.path.join(ROOT, "rust-brotli-decompressor_amd", "sharding.py"))


def build_workload(name, n_unique=None):
    """-> (label, unique streams [(compressed, raw_size, sha256)], streams per GPU).  Stream i of a job is unique[i % len(unique)]."""
    import workloads as w
    m = re.fullmatch(r"fixture:([\w.]+)x(\d+)", name)  # any of the reference's fixtures, n copies (cliff hunting)
    if m:
        return "%s x %s (reference fixture)" % (m.group(2), m.group(1)), w.fixture_streams(m.group(1)), int(m.group(2))
    m = re.fullmatch(r"recompressed:([\w.]+)q(\d+)x(\d+)", name)  # a fixture's text through the image's encoder at another quality, n copies
    if m and w.encoder_available():
        import hashlib
        (comp, size, sha), = w.fixture_streams(m.group(1))
        info, raw = load_pkg().brotli_decode(comp, size)   # (the product decodes the fixture: checked against the manifest's hash)
        if info.result != 1 or hashlib.sha256(raw).hexdigest() != sha:
            raise SystemExit("fixture %s did not decode" % m.group(1))
        q = int(m.group(2))
        return ("%s x %s recompressed with brotli -q%d, wbits 22 (real text: literals without context at this quality, a word of the static dictionary every few dozen commands; copies of ONE stream)"
                % (m.group(3), m.group(1), q)), [(w.brotli_compress(raw, q, 22), size, sha)], int(m.group(3))
    m = re.fullmatch(r"alice29x(\d+)", name)
    if m or not w.encoder_available():
        n = int(m.group(1)) if m else 1024
        return "%d x alice29.txt.compressed (reference fixture, wbits 22)" % n, w.fixture_streams("alice29.txt.compressed"), n
    m = re.fullmatch(r"longbackrefmix_(\d+)", name)   # one 64 MiB stream among n - 1 of 1 MiB: a pool of blocks
    if m:
        n = int(m.group(1))
        big = w.make_streams("long_backref", 1, 64 << 20, 1000)
        small = w.make_streams("long_backref", 8, 1 << 20, 3000)
        return ("1 x 65536 KiB + %d x 1024 KiB streams, wbits 22, brotli -q5, long back-references" % (n - 1)), big + [small[i % 8] for i in range(n - 1)], n
    m = re.fullmatch(r"(longbackref|highentropy|surveymix)(?:q(\d+))?_(\d+)x(\d+)(KiB|MiB)", name)
    if not m:
        raise SystemExit("unknown workload " + name)
    kind, quality, n, size = m.group(1), int(m.group(2) or 5), int(m.group(3)), int(m.group(4)) << (10 if m.group(5) == "KiB" else 20)
    nu = min(n, n_unique if n_unique else int(os.environ.get("BROTLI_BENCH_UNIQUE", "256")))
    seed0 = {"longbackref": 1000, "highentropy": 2000, "surveymix": 4000}[kind] if (n % 256, size) == (0, 4 << 20) else 3000  # (512 x 4 MiB: the headline's streams, twice)
    u = w.make_streams({"longbackref": "long_backref", "highentropy": "high_entropy", "surveymix": "survey_mix"}[kind], nu, size, seed0, quality=quality)
    what = {"longbackref": "long back-references", "highentropy": "high-entropy literals",
            "surveymix": "long back-references behind a seed of a quarter of the stream (the make-up of SURVEY 8a1's prototype)"}[kind]
    if (n, size) == (256, 4 << 20):
        label = "1 GiB batch: 256 x 4 MiB streams, wbits 22, brotli -q%d, %s (%d distinct streams)" % (quality, what, nu)
    else:
        label = "%d x %d KiB streams, wbits 22, brotli -q%d, %s (%d distinct streams)" % (n, size >> 10, quality, what, nu)
    return label, u, n


class DeviceJob:
    """A list of streams resident in HBM (own compressed copy and own output buffer each) and the batch object that decodes it."""

    def __init__(self, pkg, torch, dev, unique, indices):
        self.pkg, self.torch, self.unique, self.indices = pkg, torch, unique, list(indices)
        n = len(self.indices)
        self.in_stride = max((len(c) + 255) // 256 * 256 for c, _, _ in unique)
        self.out_stride = max((sz + 255) // 256 * 256 for _, sz, _ in unique)
        self.d_in = torch.zeros(max(1, n) * self.in_stride, dtype=torch.uint8, device=dev)
        self.d_out = torch.zeros(max(1, n) * self.out_stride, dtype=torch.uint8, device=dev)
        staged = {}
        self.sizes, self.caps = [], []
        for j, i in enumerate(self.indices):
            u = i % len(unique)
            c, sz, _ = unique[u]
            if u not in staged:
                staged[u] = torch.frombuffer(bytearray(c), dtype=torch.uint8).to(dev)
            self.d_in[j * self.in_stride: j * self.in_stride + len(c)] = staged[u]
            self.sizes.append(len(c))
            self.caps.append(sz)
        self.in_ptrs = [self.d_in.data_ptr() + j * self.in_stride for j in range(n)]
        self.out_ptrs = [self.d_out.data_ptr() + j * self.out_stride for j in range(n)]
        torch.cuda.synchronize()
        self.batch = pkg.Batch(max(1, n), lds_arena_bytes=int(os.environ.get("BROTLI_BENCH_LDS_ARENA", "0")))
        self.stream = torch.cuda.current_stream().cuda_stream
        self.comp_total = sum(self.sizes)
        self.raw_total = sum(self.caps)

    def decode_and_check(self):
        """First decode + bit-exact check against the regenerated raw data (SHA-256 per stream) -> status rows"""
        if not self.indices:
            self.second_pass = 0
            self.probe_ms = 0.0
            self.blocks_per_stream = 1
            self.pool = False
            self.num_commands = self.engine_commands = 0
            return []
        self.batch.decode_device(self.in_ptrs, self.sizes, self.out_ptrs, self.caps, self.pkg.FLAG_LARGE_WINDOW, self.stream)
        res = self.batch.wait()
        self.probe_ms = self.batch.last_probe_ms()   # (batch.h: what the call spent asking the device what kind the streams are; relaunches never probe)
        # The timed region re-runs the (first-pass) kernel only: it is the whole job as long as no stream needed the
        # second, large-arena launch (reported so that it cannot go unnoticed; then the whole submit + wait is timed).
        self.second_pass = self.batch.last_second_pass_count()
        self.blocks_per_stream = self.batch.last_gang()   # (1, or 2 / 4 / 8: a gang of blocks on every stream of a small batch)
        self.pool = self.batch.last_pool()                # (blocks without a stream of their own help the largest stream still being decoded)
        self.num_commands = sum(int(r.num_commands) for r in res)
        self.engine_commands = sum(int(r.engine_commands) for r in res)
        bad = [j for j, r in enumerate(res) if r.result != 1 or r.decoded_size != self.caps[j]]
        if bad:
            raise SystemExit("decode failed: stream %d result %d error %d" % (self.indices[bad[0]], res[bad[0]].result, res[bad[0]].error_code))
        host = self.d_out.cpu().numpy()
        for j, i in enumerate(self.indices):
            _, sz, sha = self.unique[i % len(self.unique)]
            if hashlib.sha256(host[j * self.out_stride: j * self.out_stride + sz].tobytes()).hexdigest() != sha:
                raise SystemExit("stream %d is not bit-exact" % i)
        return [[r.result, r.error_code, r.decoded_size, r.consumed] for r in res]

    def poison(self):
        """every output byte overwritten with a pattern no stream decodes to: what is hashed afterwards was written since"""
        self.d_out.fill_(0xA5)
        self.torch.cuda.synchronize()

    def verify(self):
        host = self.d_out.cpu().numpy()
        for j, i in enumerate(self.indices):
            _, sz, sha = self.unique[i % len(self.unique)]
            if hashlib.sha256(host[j * self.out_stride: j * self.out_stride + sz].tobytes()).hexdigest() != sha:
                raise SystemExit("stream %d is not bit-exact after the timed steps" % i)

    def step(self):
        if not self.indices:
            return 0.0
        if self.second_pass:
            ts = time.perf_counter()
            self.batch.decode_device(self.in_ptrs, self.sizes, self.out_ptrs, self.caps, self.pkg.FLAG_LARGE_WINDOW, self.stream)
            self.batch.wait()
            return (time.perf_counter() - ts) * 1e3
        self.batch.relaunch(self.stream)
        # HIP events recorded around the launch on the launch stream; reading them waits for this step only
        return self.batch.last_kernel_ms()

    def call_ms(self):
        """one whole call as a caller makes it -- BrotliAmdBatchDecodeDevice + BrotliAmdBatchWait, host clock -- beside the kernel time of the
        timed relaunches (the probe's answer is kept with the batch object: its cost, paid once, is `probe_ms`)"""
        if not self.indices:
            return 0.0
        self.torch.cuda.synchronize()
        ts = time.perf_counter()
        self.batch.decode_device(self.in_ptrs, self.sizes, self.out_ptrs, self.caps, self.pkg.FLAG_LARGE_WINDOW, self.stream)
        self.batch.wait()
        return (time.perf_counter() - ts) * 1e3

    def close(self):
        self.batch.close()


def pmc_traffic(workload):
    """HBM bytes per launch from the PMC passes of the newest committed profile of this workload (separate rocprofv3 runs of the same
    command: FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, tools/collect_profiles.sh) -> (bytes or None, file)"""
    import glob
    for prof in sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_r*.json")), reverse=True):
        try:
            pmc = json.load(open(prof))
            if pmc["workload"] == workload and pmc.get("traffic_bytes_per_launch"):
                return pmc["traffic_bytes_per_launch"], "profiles/" + os.path.basename(prof)
        except (OSError, KeyError, ValueError):
            pass
    return None, None


def roofline(comp, raw, kernel_ms, traffic=None):
    achieved = (comp + raw) / (kernel_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
            "kernel": "brotli_amd_decode_kernel", "kernel_ms": round(kernel_ms, 3),
            "decompressed_frac": round(raw / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}


def command_rates(job, seconds_per_step):
    """what the decode costs is per command: the honest unit beside MB/s"""
    c = max(1, job.num_commands)
    return {"commands": job.num_commands, "commands_per_s": round(job.num_commands / seconds_per_step, 1), "bytes_per_command": round(job.raw_total / c, 1),
            "engine_commands_share": round(job.engine_commands / c, 4), "compression_ratio": round(job.raw_total / max(1, job.comp_total), 2)}


def time_single_gpu(pkg, torch, dev, name, steps, warmup, n_unique=None, cpu_budget_s=3.0):
    """One configuration on this GPU, bit-exact checked -> extra_configs entry"""
    label, unique, n = build_workload(name, n_unique)
    job = DeviceJob(pkg, torch, dev, unique, range(n))
    job.decode_and_check()
    for _ in range(warmup):
        job.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms = [job.step() for _ in range(steps)]
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kms = sum(ms) / len(ms)
    traffic, _ = pmc_traffic(name)
    rl = roofline(job.comp_total, job.raw_total, kms, traffic)
    cr = command_rates(job, elapsed / steps)
    # one short record a leg (the driver keeps the last 8 KB of this line): ids and units in `extra_legend`
    out = {"id": name, "n": n, "D": job.raw_total, "C": job.comp_total, "MBps": round(job.raw_total * steps / elapsed / 1e6, 1), "kernel_ms": rl["kernel_ms"],
           "frac": rl["frac"], "dfrac": rl["decompressed_frac"], "traffic": traffic, "Mcmd_s": round(cr["commands_per_s"] / 1e6, 1), "B_cmd": cr["bytes_per_command"],
           "eng": cr["engine_commands_share"], "pass2": job.second_pass, "cus": job.blocks_per_stream}
    if job.pool:
        out["pool"] = 1
    out["call_ms"] = round(min(job.call_ms() for _ in range(2)), 3)
    if job.probe_ms:
        out["probe_ms"] = round(job.probe_ms, 3)
    job.close()
    if cpu_budget_s:  # the CPU path beside it: the oracle on all host threads and on one, a bounded sample of this leg's streams
        try:
            cb = cpu_baseline(unique, cpu_budget_s, proxy=False)
            out["cpu"] = [cb["value"], cb["value_1thread"], cb["cores"]]
        except Exception as ex:  # noqa: BLE001 -- a failing CPU leg must not hide the GPU number
            out["cpu"] = str(ex)[:60]
    return out


def cpu_baseline(unique, budget_s=10.0, proxy=True):
    """CPU legs on a bounded sample of the same streams: the oracle on all host threads (one stream per thread), on one
    thread, and Google's libbrotlidec on one thread when it can be loaded."""
    import oracle_lib as oracle
    L = oracle.lib()
    cores = os.cpu_count() or 1
    # one thread a stream; small streams several a thread (about 4 MiB of output each: a thread's start-up is not the measurement);
    # at most 2 GiB of output buffers
    avg_raw = max(1, sum(sz for _, sz, _ in unique) // len(unique)); max_raw = max(sz for _, sz, _ in unique)
    n = cores * max(1, (4 << 20) // avg_raw)   # (4 MiB streams: one a thread, the workload's own shape)
    n = max(1, min(n, 4096, (2 << 30) // (max_raw + 64)))
    sample = (unique * ((n + len(unique) - 1) // len(unique)))[:n]
    cores = min(cores, n)
    ins = [ctypes.create_string_buffer(c, len(c)) for c, _, _ in sample]
    outs = [ctypes.create_string_buffer(sz + 64) for _, sz, _ in sample]
    a_in = (ctypes.c_void_p * n)(*[ctypes.addressof(b) for b in ins])
    a_is = (ctypes.c_size_t * n)(*[len(c) for c, _, _ in sample])
    a_out = (ctypes.c_void_p * n)(*[ctypes.addressof(b) for b in outs])
    a_oc = (ctypes.c_size_t * n)(*[sz + 64 for _, sz, _ in sample])
    infos = (oracle.OracleInfo * n)()
    L.brotli_oracle_decode_batch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
    total = sum(sz for _, sz, _ in sample)
    best, reps, t_start = None, 0, time.time()
    while reps < 3 or (time.time() - t_start < budget_s and reps < 50):
        t = time.time()
        L.brotli_oracle_decode_batch(n, a_in, a_is, a_out, a_oc, 1, cores, infos)
        dt = time.time() - t
        best = dt if best is None else min(best, dt)
        reps += 1
    assert all(i.result == 1 for i in infos)
    # one thread: a few streams, one after the other
    k1 = min(n, 4)
    best1, t_start = None, time.time()
    for _ in range(5):
        t = time.time()
        L.brotli_oracle_decode_batch(k1, a_in, a_is, a_out, a_oc, 1, 1, infos)
        dt = time.time() - t
        best1 = dt if best1 is None else min(best1, dt)
        if time.time() - t_start > 4.0:
            break
    total1 = sum(sz for _, sz, _ in sample[:k1])
    out = {"value": round(total / best / 1e6, 1), "unit": "MB/s decompressed", "cores": cores, "kind": "port",
           "value_1thread": round(total1 / best1 / 1e6, 1),
           "sample": "%d streams of the workload (%.0f MiB), one a thread at a time, best of %d passes; 1 thread: %d streams; oracle/ (the Rust reference cannot be built here)" % (n, total / 2**20, reps, k1)}
    if not proxy:
        return out
    try:
        import libbrotli_ref as ref
        if ref.available():
            # Google's libbrotlidec (the C decoder the reference is a port of) through the oracle library's pthread harness: its
            # one-shot entry point called from real threads (the interpreter is not in the timed region)
            fn = ctypes.cast(ref._dec.BrotliDecoderDecompress, ctypes.c_void_p)
            L.brotli_oracle_run_batch_fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
            ok = (ctypes.c_int * n)()
            bestn = bestp = None
            for _ in range(3):
                t = time.time()
                L.brotli_oracle_run_batch_fn(fn, n, a_in, a_is, a_out, a_oc, cores, ok)
                dt = time.time() - t
                bestn = dt if bestn is None else min(bestn, dt)
            assert all(v == 1 for v in ok)
            for _ in range(3):
                t = time.time()
                L.brotli_oracle_run_batch_fn(fn, k1, a_in, a_is, a_out, a_oc, 1, ok)
                dt = time.time() - t
                bestp = dt if bestp is None else min(bestp, dt)
            out["libbrotlidec_proxy"] = {"value": round(total / bestn / 1e6, 1), "value_1thread": round(total1 / bestp / 1e6, 1), "cores": cores, "unit": "MB/s decompressed",
                                         "note": "Google libbrotlidec 1.0.9 from %d pthreads (1 thread: %d streams), best of 3: a proxy for the reference, which was not run" % (cores, k1)}
    except Exception as e:  # noqa: BLE001 -- the proxy is optional
        out["libbrotlidec_proxy"] = {"error": str(e)[:100]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("BROTLI_BENCH_WORKLOAD", "longbackref_256x4MiB"),
                    help="longbackref_256x4MiB (default, the metric's configuration), highentropy_256x4MiB, alice29x1024, "
                         "fixture:<name>x<n>, or <longbackref|highentropy>_<streams>x<size><KiB|MiB> for occupancy sweeps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs legs (C2, C4, C3 as one stream)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the decode path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")  # RCCL over xGMI
    pkg = load_pkg()
    sharding = load_sharding()

    # rank 0 decides the workload and broadcasts its name; the streams are a function of (workload, stream index)
    desc = [args.workload]
    if world > 1:
        dist.broadcast_object_list(desc, src=0)
    label, unique, per_gpu = build_workload(desc[0])
    n_total = per_gpu * world
    # descriptor table of the whole job on rank 0 -> every rank (RCCL broadcast); LPT partition; this rank's part
    table = np.array([[len(unique[i % len(unique)][0]), unique[i % len(unique)][1], unique[i % len(unique)][1]] for i in range(n_total)],
                     dtype=np.int64).reshape(-1, sharding.DESC_COLS)
    table = sharding.broadcast_descriptors(table if rank == 0 else None, 0, dev if world > 1 else "cpu")
    mine = sharding.lpt_partition(table[:, 2], world)[rank]

    job = DeviceJob(pkg, torch, dev, unique, mine)
    status_rows = job.decode_and_check()

    for _ in range(args.warmup):
        job.step()
    job.poison()  # (outside the timed region) the outputs hashed behind the timed steps are the timed steps' own
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = [job.step() for _ in range(args.steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    job.verify()

    # per-stream status words back to every rank (the gather of sharding.py), the step time as the maximum over ranks
    status = sharding.gather_status(np.array(status_rows, dtype=np.int64).reshape(-1, sharding.STATUS_COLS), mine, n_total, dev if world > 1 else "cpu")
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        assert int((status[:, 0] == 1).sum()) == n_total, "a stream of the job did not decode"
        elapsed_max = float(t.item())
        total_raw = float(status[:, 2].sum())
        value = total_raw * args.steps / elapsed_max / 1e6
        mean_kernel_ms = sum(kernel_ms) / len(kernel_ms)
        # HBM bytes per launch from the PMC passes of the committed profile (same command, separate rocprofv3 runs)
        traffic, traffic_source = pmc_traffic(desc[0])
        if traffic_source:
            traffic_source += ": separate rocprofv3 --pmc passes of this command (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE), not measured in this run"
        out = {
            "metric": "decompressed MB/s (bit-exact vs reference fixtures; HIP decode kernel, inputs resident in HBM)",
            "value": round(value, 1), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": label, "streams_per_gpu": per_gpu, "streams_total": n_total, "decompressed_bytes_per_gpu": job.raw_total,
                       "compressed_bytes_per_gpu": job.comp_total,
                       "parallelism": "independent streams, LPT partition over %d GPU(s) (sharding.py), no data-path collective" % world,
                       "second_pass_streams": job.second_pass, "blocks_per_stream": job.blocks_per_stream, "pool_of_blocks": bool(job.pool)},
            "roofline": roofline(job.comp_total, job.raw_total, mean_kernel_ms, traffic),
        }
        out["roofline"]["traffic_source"] = traffic_source
        if world == 1:
            out.update(command_rates(job, elapsed_max / args.steps))
        out["config"]["outputs_poisoned_before_and_hashed_after_the_timed_steps"] = True
    job.close()
    if rank == 0 and world == 1:
        if not args.no_cpu_baseline:  # (a host-side baseline: rank 0 at N = 1 only)
            out["cpu_baseline"] = cpu_baseline(unique)
            try:  # the same batch from pageable host memory and back (BrotliAmdBatchDecodeHost): PCIe-inclusive, never the `value`
                datas = [np.frombuffer(unique[i % len(unique)][0], dtype=np.uint8).copy() for i in range(per_gpu)]
                caps_h = [unique[i % len(unique)][1] for i in range(per_gpu)]
                outs_h = [np.empty(c, dtype=np.uint8) for c in caps_h]
                hb = pkg.Batch(per_gpu)
                args_h = ([a.ctypes.data for a in datas], [a.size for a in datas], [a.ctypes.data for a in outs_h], caps_h, pkg.FLAG_LARGE_WINDOW)
                hb.decode_host_raw(*args_h)  # (staging buffers allocated and pinned)
                best_h = None
                for _ in range(3):
                    for a in outs_h:
                        a[:64] = 0xA5
                    th = time.perf_counter()
                    res_h = hb.decode_host_raw(*args_h)
                    dt_h = time.perf_counter() - th
                    best_h = dt_h if best_h is None else min(best_h, dt_h)
                hb.close()
                assert all(r.result == 1 for r in res_h)
                for i in (0, per_gpu // 2, per_gpu - 1):
                    assert hashlib.sha256(outs_h[i].tobytes()).hexdigest() == unique[i % len(unique)][2]
                out["host_buffers"] = {"value": round(sum(caps_h) / best_h / 1e6, 1), "unit": "MB/s decompressed", "seconds": round(best_h, 4),
                                       "note": "BrotliAmdBatchDecodeHost, caller's pageable buffers both ways (pinned staging inside, transfers in pieces side by side with the host's copies); best of 3"}
                del outs_h, datas
            except Exception as ex:  # noqa: BLE001 -- an optional leg
                out["host_buffers"] = {"error": str(ex)[:120]}
        if not args.no_extra and desc[0] == "longbackref_256x4MiB":
            import workloads as w
            extra = []
            legs = []
            if w.encoder_available():
                legs += [("longbackref_512x4MiB", 3, None), ("longbackrefq9_256x4MiB", 5, None), ("longbackref_1024x1MiB", 5, None), ("longbackref_4096x1MiB", 3, None),
                         ("recompressed:lcet10.txt.compressedq5x1024", 3, None), ("recompressed:lcet10.txt.compressedq5x256", 3, None), ("recompressed:mapsdatazrh.compressedq5x1024", 3, None),
                         ("surveymix_256x4MiB", 5, None), ("longbackref_32x4MiB", 5, None), ("longbackrefmix_200", 3, None), ("longbackref_1x64MiB", 3, 1)]
                if os.environ.get("BROTLI_BENCH_NO_1GIB") is None:
                    legs.append(("longbackref_1x1024MiB", 1, 1))  # BASELINE config 3 as written: ONE stream of 1 GiB
                legs.append(("highentropy_256x4MiB", 5, None))
            legs.append(("alice29x4096", 3, None))    # (sixteen streams a CU: four-wave blocks that take them off the queue, the device's choice since round 6)
            legs.append(("alice29x1024", 10, None))   # (the BASELINE configurations last: the end of the line is what a truncated copy keeps)
            for name, steps, nu in legs:
                try:
                    extra.append(time_single_gpu(pkg, torch, dev, name, steps, 0 if name == "longbackref_1x1024MiB" else 1, nu, cpu_budget_s=0.0 if args.no_cpu_baseline else 3.0))
                except SystemExit as ex:  # a failing leg must not hide the headline
                    extra.append({"id": name, "error": str(ex)[:100]})
            out["extra_legend"] = ("id = bench.py --workload name (alice29x1024 = BASELINE config 2; alice29x4096 = the same fixture sixteen a CU; highentropy_256x4MiB = config 4; longbackref_1x1024MiB = config 3 as written, ONE 1 GiB stream, "
                                   "1x64MiB the same at 64 MiB; surveymix = SURVEY 8(a1)'s make-up, a quarter of each stream Zipf seed; 512x4MiB = the metric's streams twice; q9 = the metric's data at -q9; "
                                   "1024x1MiB = its make-up in 1 MiB streams (four a CU: engine blocks, the device's choice), 4096x1MiB = sixteen a CU (one-wave blocks: streams in flight); recompressed:lcet10 = real text at -q5, 256 / 1024 copies -- 1024: four a CU on a wave each with the command records, the device's choice since round 6; recompressed:mapsdatazrh = the reference's map-tile fixture at -q5, 1024 copies: neither text nor long copies, one-wave blocks); n streams, D / C bytes out / in, MBps decompressed whole job, "
                                   "frac = (C+D)/t/8 TB/s, dfrac = D/t/8 TB/s, traffic = HBM bytes a launch from profiles/pmc_r*_<id>.json (null: no PMC pass committed), Mcmd_s = million commands/s, "
                                   "B_cmd bytes a command, eng = share of commands a command engine took, pass2 = streams that needed a second launch, cus = blocks (CUs) that worked on each stream (batches of up to half the CUs' streams: gangs of 8 / 4 / 2; 32x4MiB = 32 of the metric's streams, eight blocks each; mix_200 = one 64 MiB stream among 199 of 1 MiB, a POOL launch: blocks without a stream of their own help the largest stream still being decoded), cpu = [oracle MB/s on all host threads, on 1 thread, threads]")
            out["extra_configs"] = extra
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
