"""world_size-2 test of the data-parallel path on CPU (gloo): the sharding rules of
speech_recognition_amd.parallel, driven with the oracle network standing in for the HIP replica
(the device kernels themselves are covered by the -m gpu tests, including row_offset / loss_batch).
What must hold: the all-reduced gradient equals the sum of the shard gradients computed with global
dropout rows and 1/(B*W) scaling; replicas that start equal stay bit-identical after the step; rank
sampler seeds differ."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oracle import layers as OL
    from oracle.net import TimeSlicedAttentionNet
    from speech_recognition_amd import parallel
    parallel.init_from_env(backend="gloo")
    assert parallel.active() and parallel.world_size() == world and parallel.rank() == rank
    net = TimeSlicedAttentionNet(num_classes=12, input_size=8000, dtype=np.float64)   # 0.5 s clips: fast on CPU
    rng = np.random.RandomState(0)
    gb = 4                                                           # global batch
    x = rng.randn(gb, 8000) * 0.1
    y = np.eye(12)[rng.randint(0, 12, gb)]
    lo, hi = parallel.shard_rows(gb)
    loss, p, grads, _ = net.loss_and_grads(x[lo:hi], y[lo:hi], seed=5, step=0, drop_offset=lo, loss_scale_B=gb)
    flat = torch.from_numpy(np.concatenate([g.reshape(-1) for g in grads.values()]))
    local = flat.clone()
    parallel.allreduce_grads(flat)
    # the split form (KWS_ALLREDUCE_SPLIT: late layers' slice started early, early layers' slice afterwards) must give the
    # bits of the one-buffer all-reduce
    flat2 = local.clone()
    off = 12345
    h = parallel.allreduce_begin(flat2[off:])
    parallel.allreduce_grads(flat2[:off])
    parallel.allreduce_wait(h)
    assert torch.equal(flat2, flat)
    # what fit_generator LOGS per epoch: the global batch's sums, the same on every rank
    assert parallel.allreduce_sums([1.5 + rank, 10.0 * (rank + 1), 2.0]) == [4.0, 30.0, 4.0]
    # optimizer step on the reduced gradient: replicas stay identical
    p0 = np.concatenate([v.reshape(-1).astype(np.float64) for v in net.params.values()])
    new_p, _ = OL.rmsprop_step(p0, flat.numpy(), np.zeros_like(p0), 1e-3)
    np.save(os.path.join(out_dir, "r%d.npy" % rank),
            np.stack([local.numpy(), flat.numpy(), new_p]))
    np.save(os.path.join(out_dir, "seed%d.npy" % rank), np.array([parallel.rank_seed(1234), lo, hi]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_gradient_allreduce(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    np.testing.assert_allclose(r0[1], r0[0] + r1[0], rtol=1e-12, atol=1e-15)   # sum of shard gradients
    assert np.array_equal(r0[1], r1[1])                                        # same reduced gradient
    assert np.array_equal(r0[2], r1[2])                                        # replicas stay in sync
    assert not np.allclose(r0[0], r1[0])                                       # shards really differ
    s0, s1 = np.load(tmp_path / "seed0.npy"), np.load(tmp_path / "seed1.npy")
    assert list(s0) == [1234, 0, 2] and list(s1) == [1235, 2, 4]


def _worker8(rank, world, port, out_dir):
    """World-8 rehearsal of the host-side N > 1 paths (the shapes of BASELINE configs[3] / configs[4]: global batch 8192, the
    158,538-clip test set).  No GPU: 8 ranks on one card are not allowed on this pool, so the pieces that do not need the device
    run here over gloo - row sharding, sampler seeds, the test-set partition, the gradient-sized all-reduce, the per-rank gathers
    and epoch sums of bench.py / fit_generator."""
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    import bench
    from speech_recognition_amd import parallel, tta
    parallel.init_from_env(backend="gloo")
    assert parallel.active() and parallel.world_size() == world and parallel.rank() == rank
    lo, hi = parallel.shard_rows(8192)                               # configs[3]: 1024 clips per rank
    t_lo, t_hi = tta.shard_range(158538)                             # configs[4]: make_submission.py's test set, range-partitioned
    flat = torch.full((1191436,), float(rank + 1))                   # the flat gradient buffer of the headline net
    parallel.allreduce_grads(flat)
    assert float(flat.min()) == float(flat.max()) == world * (world + 1) / 2.0
    flat2 = torch.full((1191436,), float(rank + 1))                  # the split form: late slice begun first, early slice after
    h = parallel.allreduce_begin(flat2[700000:])
    parallel.allreduce_grads(flat2[:700000])
    parallel.allreduce_wait(h)
    assert torch.equal(flat2, flat)
    sums = parallel.allreduce_sums([1.0, float(rank), 0.5])          # fit_generator's global-batch epoch metrics
    assert sums == [float(world), float(sum(range(world))), 0.5 * world]
    per_rank = bench.gather_per_rank(dist, 4.0 + 0.25 * rank, world, torch.device("cpu"))     # bench.py's per_rank_ms
    seeds = bench.gather_per_rank(dist, float(parallel.rank_seed(1234)), world, torch.device("cpu"))
    np.save(os.path.join(out_dir, "w8_r%d.npy" % rank), np.array([lo, hi, t_lo, t_hi] + per_rank + seeds))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_eight_rank_gloo_rehearsal(tmp_path):
    world, port = 8, _free_port()
    mp.spawn(_worker8, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rows = [np.load(tmp_path / ("w8_r%d.npy" % r)) for r in range(world)]
    assert [(int(r[0]), int(r[1])) for r in rows] == [(1024 * i, 1024 * (i + 1)) for i in range(world)]
    # the test set is covered exactly once, in rank order, by contiguous ranges
    t = [(int(r[2]), int(r[3])) for r in rows]
    assert t[0][0] == 0 and t[-1][1] == 158538 and all(t[i][1] == t[i + 1][0] for i in range(world - 1))
    assert all(b > a for a, b in t) and sum(b - a for a, b in t) == 158538
    for r in rows:                                                   # every rank holds every rank's time and seed, in rank order
        assert list(r[4:4 + world]) == [4.0 + 0.25 * i for i in range(world)]
        assert list(r[4 + world:]) == [1234.0 + i for i in range(world)] and len(set(r[4 + world:])) == world


def test_single_process_is_a_noop():
    from speech_recognition_amd import parallel
    t = torch.ones(4)
    assert parallel.allreduce_grads(t) is t and parallel.world_size() == 1 and parallel.shard_rows(8) == (0, 8)
    assert parallel.allreduce_sums([1.0, 2.5]) == [1.0, 2.5]
