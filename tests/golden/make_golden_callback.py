#!/usr/bin/env python
"""Golden vectors for the validation callback, produced BY THE REFERENCE's own callbacks.py.

Build container only (needs /root/reference):   python tests/golden/make_golden_callback.py

The reference's `ConfusionMatrixCallback` / `log_loss` (callbacks.py:6-83) are imported unmodified and driven with a
fake model whose `predict` returns seeded probabilities.  Its two third-party imports are not installed here:
  * keras.callbacks.Callback  -> an empty base class (the reference only inherits from it);
  * pandas_ml.ConfusionMatrix -> the few lines of pandas_ml 0.5 the callback touches, restated with pandas (installed):
      `_df_confusion = pd.crosstab(y_true, y_pred)` re-indexed on both axes to the SORTED UNION of the labels seen
      (rows = actual, columns = predicted, missing cells 0), `to_dataframe()` returns it.
    This stand-in is the one assumption of the fixture; everything else - log_loss clipping, per-class accuracy =
    diag / row sum with 0.0 for empty rows, the float32 mean, the wanted/_unknown_ folding, the epoch line format and the
    keys injected into `logs` - is the reference's code running.
Stored: the inputs (labels as class indices, probabilities) and the observed `logs` values / first lines of the two
text files.  No reference source text is stored.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import pandas as pd

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def install_stubs():
    keras = types.ModuleType('keras')
    kcb = types.ModuleType('keras.callbacks')

    class Callback(object):
        pass
    kcb.Callback = Callback
    keras.callbacks = kcb
    sys.modules['keras'] = keras
    sys.modules['keras.callbacks'] = kcb
    pml = types.ModuleType('pandas_ml')

    class ConfusionMatrix(object):
        def __init__(self, y_true, y_pred):
            yt = pd.Series(list(y_true), name='Actual')
            yp = pd.Series(list(y_pred), name='Predicted')
            df = pd.crosstab(yt, yp)
            idx = sorted(set(yt) | set(yp))
            self._df_confusion = df.reindex(index=idx, columns=idx, fill_value=0)

        def to_dataframe(self):
            return self._df_confusion
    pml.ConfusionMatrix = ConfusionMatrix
    sys.modules['pandas_ml'] = pml


class FakeModel(object):
    def __init__(self, preds):
        self.preds = list(preds)

    def predict(self, X):
        return self.preds.pop(0)


def case(name, words, wanted, n_batches, batch, seed, drop_classes=()):
    """words: label list (index = class id).  drop_classes never occur as truth (empty confusion rows)."""
    rng = np.random.RandomState(seed)
    C = len(words)
    ys, ps = [], []
    for _ in range(n_batches):
        lab = rng.randint(0, C, batch)
        for d in drop_classes:
            lab[lab == d] = (d + 1) % C
        logits = rng.randn(batch, C) * 1.5
        logits[np.arange(batch), lab] += 2.0 * (rng.rand(batch) < 0.7)
        p = np.exp(logits - logits.max(1, keepdims=True))
        p = (p / p.sum(1, keepdims=True)).astype(np.float32)
        p[0, :] = 0.0
        p[0, lab[0]] = 1.0            # an exact 0/1 row: exercises the 1e-12 clip of log_loss
        ys.append(np.eye(C, dtype=np.float32)[lab])
        ps.append(p)
    return dict(name=name, words=words, wanted=wanted, y_true=[y.argmax(1).tolist() for y in ys],
                y_pred=[p.tolist() for p in ps]), ys, ps


def main():
    install_stubs()
    sys.path.insert(0, REF)
    import callbacks as ref_cb
    out = []
    specs = [
        ('twelve', ['_silence_', '_unknown_', 'yes', 'no', 'up', 'down', 'left', 'right', 'on', 'off', 'stop', 'go'],
         ['yes', 'no', 'up', 'down', 'left', 'right', 'on', 'off', 'stop', 'go'], 3, 16, 11, ()),
        ('thirty_two_folded', ['_silence_', '_unknown_'] + ['w%02d' % i for i in range(30)],
         ['w00', 'w03', 'w07', 'w11', 'w12'], 2, 48, 12, ()),
        ('missing_rows', ['_silence_', '_unknown_', 'a', 'b', 'c', 'd'], ['a', 'b'], 2, 10, 13, (3, 5)),
    ]
    cwd = os.getcwd()
    for name, words, wanted, nb, bs, seed, drop in specs:
        rec, ys, ps = case(name, words, wanted, nb, bs, seed, drop)
        tmp = tempfile.mkdtemp()
        os.chdir(tmp)
        try:
            label2int = {w: i for i, w in enumerate(words)}
            gen = iter([(np.zeros((bs, 4)), y) for y in ys])
            cb = ref_cb.ConfusionMatrixCallback(gen, nb, wanted, words, label2int)
            cb.model = FakeModel(ps)
            logs = {}
            cb.on_epoch_end(7, logs)
            rec['logs'] = {k: float(v) for k, v in logs.items()}
            rec['logs_dtype'] = {k: type(v).__name__ for k, v in logs.items()}
            with open('confusion_matrix.txt') as f:
                rec['acc_line'] = f.read().split('\n')[:2]
            yt = np.concatenate(ys)
            yp = np.concatenate(ps)
            rec['log_loss'] = float(ref_cb.log_loss(yt, yp))
        finally:
            os.chdir(cwd)
        out.append(rec)
    with open(os.path.join(OUT, 'k8_callback.json'), 'w') as f:
        json.dump({'note': 'observed outputs of /root/reference/callbacks.py (see make_golden_callback.py)',
                   # which code produced each stored value: the reference's own lines, or the
                   # reference's lines fed by the pd.crosstab stand-in for pandas_ml.ConfusionMatrix
                   'provenance': {
                       'logs.val_loss': "reference only: callbacks.py log_loss (:6-10) on the concatenated batches (:46-56); no stand-in involved",
                       'log_loss': "reference only: callbacks.py log_loss (:6-10) called directly",
                       'logs.val_categorical_accuracy': "reference arithmetic (callbacks.py accuracy(): trace / sum of the confusion matrix, :37-43, used at :64) on the "
                                                        "matrix built by the pd.crosstab STAND-IN for pandas_ml.ConfusionMatrix",
                       'logs.val_mean_categorical_accuracy_all': "reference arithmetic (callbacks.py accuracies(): diag / row sum with 0 for empty rows, float32, used at :63; .mean() at :82) "
                                                                 "on the STAND-IN's matrix",
                       'logs.val_mean_categorical_accuracy_wanted': "reference arithmetic (callbacks.py:65-70, :83: wanted words, others folded into "
                                                                    "_unknown_) on the STAND-IN's matrix",
                       'acc_line': "reference format string (callbacks.py:71) over the two accuracies above, i.e. through the STAND-IN",
                       'logs_dtype': "types of the values the reference injected into `logs`",
                       'inputs (words, wanted, y_true, y_pred)': "seeded by this script; not reference outputs"},
                   'cases': out}, f)
    for r in out:
        print(r['name'], r['logs'], r['acc_line'])


if __name__ == '__main__':
    main()
