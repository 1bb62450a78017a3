"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" in the CPU tests).  The hot path shards by independent clips: rank r of W
trains on its own B clips per step (weak scaling), BatchNorm statistics stay local, and the ONLY
exchange is one all-reduce(sum) of the flat 1.19 M-float gradient buffer per step - the per-rank
gradient is already scaled by 1/(B*W) (loss_batch argument of kws_net_train_fwd_bwd), so the sum is
the global-batch mean.  The message is 4.8 MB: latency-bound on 7 x 153 GB/s xGMI links, so it is sent
as ONE buffer rather than per-tensor buckets.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init_from_env(backend=None):
    """Initialise the default process group from the torchrun environment (no-op for WORLD_SIZE=1)."""
    world, rank, local_rank = env_world()
    if world <= 1 or dist.is_initialized():
        return world, rank, local_rank
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        # KWS_ONE_DEVICE: the test hook - N ranks on ONE GPU, over gloo (a 1-GPU box cannot run RCCL x N)
        backend = "nccl" if torch.cuda.is_available() and not os.environ.get("KWS_ONE_DEVICE") else "gloo"
    if os.environ.get("KWS_ONE_DEVICE"):
        local_rank = 0
    kw = {}
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        kw["device_id"] = torch.device("cuda", local_rank)
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return world, rank, local_rank


def active():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size():
    return dist.get_world_size() if active() else 1


def rank():
    return dist.get_rank() if active() else 0


def allreduce_grads(flat_grads):
    """Sum the flat gradient buffer over all ranks in place (one collective per step)."""
    if active():
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
    return flat_grads


def allreduce_begin(flat_slice):
    """Start the sum of one contiguous slice of the gradient buffer (async); returns the handle for allreduce_wait.
    The collective is ordered after everything enqueued on the current stream so far, and runs on the communicator's own
    stream beside whatever the caller enqueues next."""
    if not active():
        return None
    return dist.all_reduce(flat_slice, op=dist.ReduceOp.SUM, async_op=True)


def allreduce_wait(handle):
    if handle is not None:
        handle.wait()           # NCCL/RCCL: the current stream waits for the collective; gloo: the host does


def allreduce_sums(values, device=None):
    """Sum a few host scalars over all ranks (float64); returns a list of floats.  Used for what is LOGGED: an epoch's
    loss / accuracy numerators and its sample count, so that every rank reports the global batch's metric instead of its
    own shard's (the training itself never needs it).  One small collective per epoch."""
    vals = [float(v) for v in values]
    if not active():
        return vals
    dev = device if (device is not None and dist.get_backend() == 'nccl') else 'cpu'
    t = torch.tensor(vals, dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.cpu().tolist()]


def split_block_from_env():
    """Default of Model.allreduce_split: KWS_ALLREDUCE_SPLIT=<block> all-reduces the gradients of blocks >= <block> (+ tail)
    while the earlier blocks' backward still runs.  Unset / 0 = one buffer after the backward pass (the default until an
    N-GPU run shows which wins: the 4.8 MB message is latency-bound either way; bench.py --gpus N measures both)."""
    try:
        return int(os.environ.get("KWS_ALLREDUCE_SPLIT", "0"))
    except ValueError:
        return 0


def shard_rows(global_batch):
    """[start, stop) rows of the global batch owned by this rank (dropout counter offset = start)."""
    w, r = world_size(), rank()
    per = global_batch // w
    return r * per, (r + 1) * per


def rank_seed(base_seed):
    """Sampler seed of this rank: every rank draws its own clips (seed + rank), rank 0 keeps `base_seed`
    so a 1-GPU run reproduces the reference's draw order."""
    return int(base_seed) + rank()


def broadcast_params(flat_params, src=0):
    """Make every replica start from rank `src`'s weights."""
    if active():
        dist.broadcast(flat_params, src=src)
    return flat_params
