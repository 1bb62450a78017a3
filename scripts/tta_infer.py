#!/usr/bin/env python
"""BASELINE config C5: TTA inference of a synthetic test set, sharded over the ranks of torch.distributed.run
(one process per GPU; no data-path collective).  Prints one JSON line on rank 0.
usage:  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 scripts/tta_infer.py [n_clips]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from speech_recognition_amd import parallel  # noqa: E402
from speech_recognition_amd.model import speech_model  # noqa: E402
from speech_recognition_amd.tta import predict_test_set  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    world, rank, local_rank = parallel.init_from_env()
    torch.cuda.set_device(0 if os.environ.get("KWS_ONE_DEVICE") else local_rank)
    model = speech_model('conv_1d_time_sliced_with_attention', 16000, num_classes=12)
    rng = np.random.RandomState(7)
    clips = (rng.randn(n, 16000) * 0.0774).astype(np.float32)           # the same "test set" on every rank
    predict_test_set(model, clips[:512])                                   # warm-up
    torch.cuda.synchronize()
    if parallel.active():
        parallel.dist.barrier()
    t0 = time.time()
    probs, amax = predict_test_set(model, clips, batch=4096)
    torch.cuda.synchronize()
    dt = time.time() - t0
    if rank == 0:
        print(json.dumps({"config": "C5: TTA x3 inference of %d clips sharded over %d rank(s)" % (n, world), "n_gpus": world,
                          "clips_per_s": n / dt, "seconds": dt, "checksum": float(probs.sum()), "argmax_hist": np.bincount(amax, minlength=12).tolist()}))
        np.save(os.environ.get("KWS_TTA_OUT", "/tmp/tta_probs_w%d.npy" % world), probs)
    if parallel.active():
        parallel.dist.barrier()
        parallel.dist.destroy_process_group()


if __name__ == "__main__":
    main()
