"""World-size-2 test of the batch-sharded path on CPU (gloo).  The per-rank compute is the oracle here (the HIP
kernels need a GPU); what is under test is the sharding arithmetic and the single scalar all-reduce of
tf_seq2seq_losses_amd/dist.py, i.e. everything the N > 1 bench path adds to the N = 1 path."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ctc_oracle as O


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _oracle_loss(labels, logits, label_length, logit_length, blank_index=0):
    return torch.from_numpy(O.classic_ctc_loss(labels.numpy(), logits.numpy(), label_length.numpy(),
                                                logit_length.numpy(), blank_index)).float()


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tf_seq2seq_losses_amd import dist as cdist
    inp = O.generate_ctc_loss_inputs(7, 16, 11, 6)  # 7 utterances over 2 ranks: shards of 4 and 3
    t = {k: torch.from_numpy(np.asarray(v)) for k, v in inp.items() if k != "blank_index"}
    local, s, n = cdist.sharded_loss(_oracle_loss, t["labels"], t["logits"], t["label_length"], t["logit_length"], 0)
    lo, hi = cdist.shard_bounds(7, rank, world)
    out[rank] = (lo, hi, local.numpy().copy(), float(s), float(n))
    dist.destroy_process_group()


def test_two_rank_sharding_and_scalar_allreduce():
    from tf_seq2seq_losses_amd import dist as cdist
    assert [cdist.shard_bounds(7, r, 2) for r in range(2)] == [(0, 4), (4, 7)]
    assert [cdist.shard_bounds(256, r, 8) for r in range(8)] == [(32 * r, 32 * r + 32) for r in range(8)]
    assert [cdist.shard_bounds(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    inp = O.generate_ctc_loss_inputs(7, 16, 11, 6)
    full = O.classic_ctc_loss(inp["labels"], inp["logits"], inp["label_length"], inp["logit_length"], 0)
    fin = np.isfinite(full)
    got = np.concatenate([out[0][2], out[1][2]])
    assert (out[0][0], out[0][1], out[1][0], out[1][1]) == (0, 4, 4, 7)
    assert np.array_equal(np.isfinite(got), fin) and np.abs(got[fin] - full[fin]).max() < 1e-4
    for r in range(2):
        assert abs(out[r][3] - full[fin].sum()) < 1e-3 and out[r][4] == fin.sum()


def _loop_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tf_seq2seq_losses_amd import dist as cdist
    calls = {"n": 0}

    def step():  # CPU tensors standing in for the kernel's loss[B] of this rank and step
        i = calls["n"]
        calls["n"] += 1
        loss = torch.arange(4, dtype=torch.float32) + 10.0 * i + 100.0 * rank
        if (i + rank) % 3 == 0:
            loss[1] = float("inf")  # an infeasible utterance: not summed, not counted
        return loss
    pairs = cdist.pipelined_steps(step, 7)
    # the form bench.py uses since the loss kernel accumulates the pair itself: step() hands over one of three int64[2]
    # buffers (fixed point, 2^-20), the loop only all-reduces it and hands it to `consume` before the buffer is recycled
    calls["n"] = 0
    bufs = [torch.zeros(2, dtype=torch.int64) for _ in range(3)]
    got = {}

    def step_reduced():
        i = calls["n"]
        loss = step()
        fin = torch.isfinite(loss)
        bufs[(i + 1) % 3].zero_()
        bufs[i % 3] += torch.stack([torch.round(loss[fin].double() * 1048576.0).sum().long(), fin.sum()])
        return bufs[i % 3]
    assert cdist.pipelined_steps(step_reduced, 7, reduced=True, consume=lambda i, pair: got.__setitem__(i, pair.tolist())) == []
    assert sorted(got) == list(range(7))
    for i, pr in enumerate(pairs):
        assert abs(got[i][0] / 1048576.0 - float(pr[0])) < 1e-3 and got[i][1] == int(pr[1])
    # pipeline depth 2 (bench.py --pipeline-depth 2): the pair of step i is waited for after step i+2 has been launched;
    # depth + 2 = 4 buffers in rotation, each consumed before it is recycled
    calls["n"] = 0
    bufs4 = [torch.zeros(2, dtype=torch.int64) for _ in range(4)]
    got2 = {}

    def step_reduced2():
        i = calls["n"]
        loss = step()
        fin = torch.isfinite(loss)
        bufs4[(i + 1) % 4].zero_()
        bufs4[i % 4] += torch.stack([torch.round(loss[fin].double() * 1048576.0).sum().long(), fin.sum()])
        return bufs4[i % 4]
    assert cdist.pipelined_steps(step_reduced2, 7, reduced=True, depth=2, consume=lambda i, pair: got2.__setitem__(i, pair.tolist())) == []
    assert got2 == got
    # fewer, larger collectives (bench.py --reduce-every 3): the pairs of three steps go out in one all-reduce; (depth + 2) * 3
    # rows in rotation, the last group is one step short... of nothing here: 7 = 3 + 3 + 1
    calls["n"] = 0
    for every, depth in ((3, 1), (2, 2)):
        calls["n"] = 0
        nrow = (depth + 2) * every
        rows = torch.zeros((nrow, 2), dtype=torch.int64)
        got3, sizes = {}, []

        def step_rows():
            i = calls["n"]
            loss = step()
            fin = torch.isfinite(loss)
            rows[(i + 1) % nrow].zero_()
            rows[i % nrow] += torch.stack([torch.round(loss[fin].double() * 1048576.0).sum().long(), fin.sum()])
            return rows[i % nrow]

        def view(first, n):
            a = first % nrow
            assert a + n <= nrow
            return rows[a:a + n]

        def take(i, grp):
            sizes.append(int(grp.shape[0]))
            for j in range(grp.shape[0]):
                got3[i - grp.shape[0] + 1 + j] = grp[j].tolist()
        assert cdist.pipelined_steps(step_rows, 7, reduced=True, depth=depth, every=every, group_view=view, consume=take) == []
        assert got3 == got, (every, depth)
        assert sizes == [every] * (7 // every) + ([7 % every] if 7 % every else [])
    out[rank] = [p.tolist() for p in pairs]
    dist.destroy_process_group()


def test_pipelined_all_reduce_loop_of_the_bench():
    """bench.py's N > 1 loop (dist.pipelined_steps): every step's [sum of finite losses, count] pair is all-reduced
    asynchronously and waited for one step later; all seven collectives complete and carry the right numbers."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_loop_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out[0] == out[1]
    for i, (s, n) in enumerate(out[0]):
        want_s, want_n = 0.0, 0
        for rank in range(2):
            loss = np.arange(4, dtype=np.float64) + 10.0 * i + 100.0 * rank
            if (i + rank) % 3 == 0:
                loss[1] = np.inf
            want_s += loss[np.isfinite(loss)].sum()
            want_n += int(np.isfinite(loss).sum())
        assert abs(s - want_s) < 1e-3 and n == want_n, (i, s, n, want_s, want_n)
