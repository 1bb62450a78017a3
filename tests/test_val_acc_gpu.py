"""Val-acc parity (BASELINE.json metric: "...; val-acc parity"): the reference's train/validate loop (train.py:56-75,
callbacks.py:45-83) on the device and on the oracle's CPU twin over the same batches - scripts/val_acc_parity.py."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_val_acc_parity_device_vs_cpu_oracle(repo_root):
    sys.path.insert(0, os.path.join(repo_root, "scripts"))
    import val_acc_parity
    res = val_acc_parity.run(epochs=12, steps=100, batch=64, val_batches=8, quiet=True)
    par = res["val_acc_parity"]
    print(par["device"], par["cpu"])
    # the bar: the SETTLED accuracy (median of the last three epochs) and the best epoch agree within the tolerance.  The
    # last epoch alone is reported (val_acc / val_acc_cpu) but not asserted: with Keras' BatchNorm momentum of 0.99 one late
    # epoch of either side can sit a class lower, and the torch-CPU twin is not run-to-run deterministic (1.000 / 0.980 /
    # 0.879 measured for the same batches)
    assert par["tolerance"] == 0.05 and par["validation_rows_disjoint_from_training"]
    assert abs(res["val_acc_settled"] - res["val_acc_cpu_settled"]) <= par["tolerance"], (par["device"]["val_acc"], par["cpu"]["val_acc"])
    assert abs(res["val_acc_best"] - res["val_acc_cpu_best"]) <= par["tolerance"]
    # both learned the 12-class tone task (chance = the largest class share, ~0.3 with 60 % 'unknown' draws folded in)
    assert res["val_acc_settled"] > 0.6 and res["val_acc_cpu_settled"] > 0.6
    # the training-side accuracies (same batches, same dropout masks) track each other as well
    assert abs(par["device"]["train_acc"][-1] - par["cpu"]["train_acc"][-1]) < 0.08
