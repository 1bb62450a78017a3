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


def test_single_process_is_a_noop():
    from speech_recognition_amd import parallel
    t = torch.ones(4)
    assert parallel.allreduce_grads(t) is t and parallel.world_size() == 1 and parallel.shard_rows(8) == (0, 8)
    assert parallel.allreduce_sums([1.0, 2.5]) == [1.0, 2.5]
