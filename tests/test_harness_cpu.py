"""Host logic of the SEA harness that needs no GPU: the sharded-statistics buffer with its single
all-reduce (world_size-2 gloo processes vs. one process) and the K8/K9 host arithmetic of
tools/worse_only.py against the reference's evalSEA goldens."""
import os
import random
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, load_golden
from oracle import sea_oracle as O

sys.path.insert(0, PKG)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _local_fill(stats, idx_list, preds, tgt, C, bs=3):
    """what tools/infer.py does per rank, with CPU integer tables instead of the K3 kernel"""
    for a in range(preds.shape[0]):
        for s in range(0, len(idx_list), bs):
            idx = idx_list[s:s + bs]
            p, t = preds[a, idx], tgt[idx]
            im, pm, tc = O.class_counts(p, t, C, per_image=True, mask_pred=True)
            stats.add_attack_batch(a, idx, im, pm, tc)
    for s in range(0, len(idx_list), bs):
        idx = idx_list[s:s + bs]
        stats.add_clean(*O.class_counts(preds[0, idx], tgt[idx], C, per_image=False, mask_pred=True))


def _worker(rank, world, port, preds, tgt, C, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tools.sea_shard import SeaStats, shard_indices
    st = SeaStats(preds.shape[0], tgt.shape[0], C)
    _local_fill(st, shard_indices(tgt.shape[0], rank, world), preds, tgt, C)
    st.all_reduce()
    if rank == 0:
        torch.save(st.buf, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tag", ["a", "c"])
def test_sharded_stats_two_ranks_equal_one_rank(tmp_path, tag):
    from tools.sea_shard import SeaStats, shard_indices
    g = load_golden(f"g7_evalsea_{tag}")
    preds, tgt, C = g["preds"], g["targets"], int(g["n_cls"])
    single = SeaStats(3, tgt.shape[0], C)
    _local_fill(single, shard_indices(tgt.shape[0], 0, 1), preds, tgt, C)
    out = str(tmp_path / "buf.pt")
    mp.spawn(_worker, args=(2, _free_port(), preds, tgt, C, out), nprocs=2, join=True)
    merged = torch.load(out)
    assert torch.equal(merged, single.buf)  # integer tables: bit-exact, independent of the sharding
    # and the merged tables reproduce the reference's evalSEA numbers
    from tools.worse_only import worst_acc_from_counts, worst_miou_from_tables
    st = SeaStats(3, tgt.shape[0], C)
    st.buf.copy_(merged)
    # the reference's evalSEA sees logs that eval_performance masked in place at ignored pixels (tools/infer.py:88-90)
    masked = torch.where(tgt.unsqueeze(0) == -1, torch.full_like(preds, -1), preds)
    ints, unions = O.per_image_tables(masked, tgt, C)
    assert torch.equal(st.inter.float(), ints) and torch.equal(st.union.float(), unions)
    random.seed(225)
    miou, sel, rounds = worst_miou_from_tables(st.inter, st.union)
    assert miou == O.worst_case_miou(ints, unions, rng=random.Random(225))[0]
    if not bool((tgt == -1).any()):   # no ignore labels: identical to the golden evalSEA run on the raw maps
        assert torch.equal(st.inter.float(), g["ints"]) and torch.equal(st.union.float(), g["unions"])
        assert miou == g["final_miou"]
    worst, indiv, _ = worst_acc_from_counts(st.correct, st.valid)
    ref_worst, ref_indiv, _ = O.worst_case_acc(preds, tgt, C)  # correctly aligned batches
    assert worst == pytest.approx(ref_worst, rel=1e-6)
    torch.testing.assert_close(indiv, ref_indiv, rtol=1e-6, atol=0)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_worst_case_host_arithmetic_matches_reference(tag):
    """N % bs == 0 for these goldens, so the reference's batch slicing is correct and comparable."""
    from tools.worse_only import worst_acc_from_counts, worst_miou_from_tables
    g = load_golden(f"g7_evalsea_{tag}")
    C = int(g["n_cls"])
    valid = ((g["targets"] >= 0) & (g["targets"] < C)).view(g["targets"].shape[0], -1).sum(-1)
    worst, indiv, _ = worst_acc_from_counts(g["ints"].sum(-1).long(), valid)
    assert worst == pytest.approx(g["worst_Acc"], rel=1e-6)
    torch.testing.assert_close(indiv, g["worst_Acc_indiv"], rtol=1e-6, atol=0)
    random.seed(225)
    miou, _, _ = worst_miou_from_tables(g["ints"], g["unions"])
    assert miou == g["final_miou"]
    # the global `random` stream was advanced exactly as the reference's random.shuffle calls would
    r = random.Random(225)
    _, _, rounds = O.worst_case_miou(g["ints"], g["unions"], rng=r)
    assert random.getstate()[1] == r.getstate()[1]


def test_shard_indices_cover_everything_once():
    from tools.sea_shard import shard_indices
    for n in (0, 1, 7, 16, 1449):
        for w in (1, 2, 3, 8):
            got = sorted(i for r in range(w) for i in shard_indices(n, r, w))
            assert got == list(range(n))


def test_sharded_stats_ade_size_two_ranks_equal_one_rank(tmp_path):
    """BASELINE configs[4]'s table size: C = 151 (ADE20K), a ragged 11 images over 2 ranks, ignore labels present.
    The packed int64 buffer (one all-reduce) merged over gloo equals the single-process buffer bit for bit."""
    from tools.sea_shard import SeaStats, shard_indices
    C, n = 151, 11
    g = torch.Generator().manual_seed(151)
    tgt = torch.randint(0, C, (n, 24, 20), generator=g)
    tgt[torch.rand(tgt.shape, generator=g) < 0.07] = -1
    preds = torch.where(torch.rand(3, n, 24, 20, generator=g) < 0.6, tgt.unsqueeze(0).clamp_min(0),
                        torch.randint(0, C, (3, n, 24, 20), generator=g))
    single = SeaStats(3, n, C)
    _local_fill(single, shard_indices(n, 0, 1), preds, tgt, C)
    out = str(tmp_path / "buf151.pt")
    mp.spawn(_worker, args=(2, _free_port(), preds, tgt, C, out), nprocs=2, join=True)
    merged = torch.load(out)
    assert merged.dtype == torch.int64 and torch.equal(merged, single.buf)
    assert merged.numel() >= 2 * 3 * n * C                      # the per-image (A,N,C) inter / union tables travel in it
