"""Runs an UNMODIFIED reference script (e.g. train.py) against the MI355X hot path:

    python -m speech_recognition_amd.run_script /path/to/train.py [args...]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
           -m speech_recognition_amd.run_script /path/to/train.py

`dropin/` is put in front of sys.path so the script's `import tensorflow / keras / input_data / model /
utils / callbacks / classes` resolve to the shells and mirrors of this package instead of the script's
own directory.  Under torchrun every rank runs the script on its own GPU with its own sampler seed;
gradients are all-reduced inside Model.train_on_batch / fit_generator."""
import os
import runpy
import sys


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    script = os.path.abspath(sys.argv[1])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dropin = os.path.join(root, 'dropin')
    for p in (root, dropin):
        while p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, root)
    sys.path.insert(0, dropin)
    from speech_recognition_amd import parallel
    world, rank, local_rank = parallel.init_from_env()
    import torch
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    if world > 1:
        import numpy as np
        np.random.seed(parallel.rank_seed(int(os.environ.get('KWS_SAMPLER_SEED', '1234'))))
    sys.argv = [script] + sys.argv[2:]
    # runpy puts the script's directory nowhere on sys.path when run_path is given a file with
    # run_name='__main__' and sys.path[0] already set: the drop-in modules win over same-named files there.
    runpy.run_path(script, run_name='__main__')


if __name__ == '__main__':
    main()
