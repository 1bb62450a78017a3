"""Validation callback of the reference (callbacks.py:6-83): runs the validation pass through
model.predict, derives val_loss / accuracies from a confusion matrix and INJECTS them into `logs`,
which is what drives ReduceLROnPlateau and ModelCheckpoint (train.py:56-68).  pandas_ml is replaced
by a NumPy confusion matrix with the same label ordering (sorted union of seen labels)."""
import numpy as np

from . import parallel
from .keras_api import Callback


def log_loss(y_true, y_pred, eps=1e-12):
    y_pred = np.clip(y_pred, eps, 1. - eps)
    return (-(np.sum(y_true * np.log(y_pred), axis=1))).mean()


def confusion_matrix(y_true, y_pred):
    """Rows = actual, columns = predicted, labels = sorted union (pandas_ml.ConfusionMatrix layout)."""
    labels = sorted(set(y_true) | set(y_pred))
    pos = {l: i for i, l in enumerate(labels)}
    m = np.zeros((len(labels), len(labels)), dtype=np.int64)
    for t, p in zip(y_true, y_pred):
        m[pos[t], pos[p]] += 1
    return labels, m


def _format(labels, m):
    w = max([len(str(l)) for l in labels] + [9])
    lines = ["%-*s " % (w, "Predicted") + " ".join("%*s" % (w, l) for l in labels)]
    for l, row in zip(labels, m):
        lines.append("%-*s " % (w, l) + " ".join("%*d" % (w, v) for v in row))
    return "\n".join(lines)


class ConfusionMatrixCallback(Callback):
    def __init__(self, validation_data, validation_steps, wanted_words, all_words, label2int):
        Callback.__init__(self)
        self.validation_data = validation_data
        self.validation_steps = validation_steps
        self.wanted_words = wanted_words
        self.all_words = all_words
        self.label2int = label2int
        self.int2label = {v: k for k, v in label2int.items()}
        for fn in ('confusion_matrix.txt', 'wanted_confusion_matrix.txt'):
            if self._writes():
                with open(fn, 'w'):
                    pass

    @staticmethod
    def _writes():
        """Data-parallel: rank 0 owns the two text files.  Asked at every use, not once at construction: a callback built
        before the process group exists (train.py builds its callbacks before fit_generator) must still see its rank -
        the launcher's RANK variable answers until torch.distributed does."""
        return (parallel.rank() if parallel.active() else parallel.env_world()[1]) == 0

    @staticmethod
    def accuracies(confusion_val):
        sums = confusion_val.sum(axis=1)
        diag = np.diag(confusion_val).astype(np.float64)
        return np.float32(np.where(sums > 0, diag / np.maximum(sums, 1), 0.0))

    @staticmethod
    def accuracy(confusion_val):
        return float(np.trace(confusion_val)) / confusion_val.sum()

    def on_epoch_end(self, epoch, logs=None):
        logs = logs if logs is not None else {}
        y_true, y_pred = [], []
        for _ in range(self.validation_steps):
            X_batch, y_true_batch = next(self.validation_data)
            y_pred.extend(self.model.predict(X_batch))
            y_true.extend(np.asarray(y_true_batch))
        y_true = np.float32(y_true)
        y_pred = np.float32(y_pred)
        val_loss = log_loss(y_true, y_pred)
        t = [self.int2label[i] for i in y_true.argmax(axis=-1)]
        p = [self.int2label[i] for i in y_pred.argmax(axis=-1)]
        labels, conf = confusion_matrix(t, p)
        accs, acc = self.accuracies(conf), self.accuracy(conf)
        tw = [y if y in self.wanted_words else '_unknown_' for y in t]
        pw = [y if y in self.wanted_words else '_unknown_' for y in p]
        wlabels, wconf = confusion_matrix(tw, pw)
        wanted_accs = self.accuracies(wconf)
        acc_line = "\n[%03d]: val_categorical_accuracy: %.2f, val_mean_categorical_accuracy_wanted: %.2f" % (
            epoch, acc, wanted_accs.mean())
        if self._writes():
            with open('confusion_matrix.txt', 'a') as f:
                f.write('%s\n' % acc_line)
                f.write(_format(labels, conf))
            with open('wanted_confusion_matrix.txt', 'a') as f:
                f.write('%s\n' % acc_line)
                f.write(_format(wlabels, wconf))
        logs['val_loss'] = val_loss
        logs['val_categorical_accuracy'] = acc
        logs['val_mean_categorical_accuracy_all'] = accs.mean()
        logs['val_mean_categorical_accuracy_wanted'] = wanted_accs.mean()
