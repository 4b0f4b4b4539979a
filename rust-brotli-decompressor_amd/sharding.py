"""Sharding a batch of independent .br streams over the GPUs of one node.

Streams share nothing (own window, own distance ring, own output), so the batch is a partition problem, not a
communication problem: rank 0 owns the descriptor table (one row per stream: compressed size, output capacity,
expected decompressed size or an estimate), broadcasts it, every rank takes its part by the same deterministic
greedy longest-processing-time rule, decodes it on its own GPU, and the per-stream status words are gathered
back.  Payload bytes never cross xGMI; the two collectives move a few dozen bytes per stream (RCCL when the
process group is "nccl", gloo in the CPU tests).
"""
import numpy as np
import torch
import torch.distributed as dist

DESC_COLS = 3    # compressed bytes, output capacity, weight (expected decompressed bytes)
STATUS_COLS = 4  # result, error code, decoded size, consumed bytes


def lpt_partition(weights, parts):
    """Greedy longest-processing-time: heaviest stream first, always to the least loaded part.
    Deterministic (ties by index), so every rank computes the same partition from the same table."""
    order = sorted(range(len(weights)), key=lambda i: (-int(weights[i]), i))
    loads = [0] * parts
    out = [[] for _ in range(parts)]
    for i in order:
        p = min(range(parts), key=lambda k: (loads[k], k))
        out[p].append(i)
        loads[p] += int(weights[i])
    for p in range(parts):
        out[p].sort()
    return out


def broadcast_descriptors(table, src=0, device="cpu"):
    """table: int64 array [n, DESC_COLS] on rank `src` (ignored elsewhere) -> the same array on every rank"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(table, dtype=np.int64)
    rank = dist.get_rank()
    n = torch.tensor([len(table) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src)
    t = torch.zeros((int(n.item()), DESC_COLS), dtype=torch.int64, device=device)
    if rank == src:
        t.copy_(torch.as_tensor(np.asarray(table, dtype=np.int64)).to(device))
    dist.broadcast(t, src)
    return t.cpu().numpy()


def gather_status(local_status, my_indices, n_total, device="cpu"):
    """local_status: int64 [len(my_indices), STATUS_COLS] -> full [n_total, STATUS_COLS] table on every rank"""
    full = torch.zeros((n_total, STATUS_COLS), dtype=torch.int64, device=device)
    if len(my_indices):
        full[torch.as_tensor(my_indices, dtype=torch.int64, device=device)] = torch.as_tensor(np.asarray(local_status, dtype=np.int64)).to(device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(full, op=dist.ReduceOp.SUM)  # rows are disjoint: a sum is a gather
    return full.cpu().numpy()


def decode_sharded(streams, out_caps, decode_fn, weights=None, device="cpu"):
    """streams: list of compressed byte strings, known on every rank (or only their sizes matter for planning).
    decode_fn(list of streams, list of caps) -> (status rows [k, STATUS_COLS], list of outputs) runs on this rank's
    device.  Returns (my indices, my outputs, full status table)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    table = np.array([[len(s), c, (weights[i] if weights is not None else c)] for i, (s, c) in enumerate(zip(streams, out_caps))],
                     dtype=np.int64).reshape(-1, DESC_COLS)
    table = broadcast_descriptors(table, 0, device)
    mine = lpt_partition(table[:, 2], world)[rank]
    status, outs = decode_fn([streams[i] for i in mine], [int(table[i, 1]) for i in mine])
    return mine, outs, gather_status(status, mine, len(table), device)
