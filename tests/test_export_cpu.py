"""The export / pseudo-label tools against what the reference's own scripts produced on the same synthetic
inputs (tests/golden/k7_export_tools.json <- tests/golden/make_golden_export.py): bit-exact."""
import hashlib
import importlib.util
import json
import os

import numpy as np
import pytest
from scipy.io import wavfile

from speech_recognition_amd import export
from speech_recognition_amd.classes import get_int2label

HERE = os.path.dirname(os.path.abspath(__file__))


def sha1(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def k7():
    with open(os.path.join(HERE, 'golden', 'k7_export_tools.json')) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def inputs():
    spec = importlib.util.spec_from_file_location('make_golden_export', os.path.join(HERE, 'golden', 'make_golden_export.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.export_inputs()


def test_head32to12_and_uint8_memmap_match_reference(k7, inputs, tmp_path):
    fnames, probs32, p12, conf, wavs = inputs
    p = export.head32to12_offline(probs32, get_int2label(wanted_only=False))
    assert p.dtype == np.float32 and p.shape == (len(fnames), 12)
    mm = export.write_probs_uint8_memmap(str(tmp_path / 'probs.uint8.memmap'), p)
    g = k7['convert']
    assert g['printed_unknown'] == 21                      # '_unknown_' itself + the 20 unwanted words
    for r, want in zip(g['rows'], g['values']):
        assert [int(v) for v in mm[r]] == want, r
    assert sha1(np.array(mm)) == g['sha1']
    back = np.memmap(str(tmp_path / 'probs.uint8.memmap'), dtype='uint8', mode='r', shape=p.shape)
    assert np.array_equal(back, np.array(mm))


def test_pseudo_label_tree_matches_reference(k7, inputs, tmp_path):
    fnames, probs32, p12, conf, wavs = inputs
    src = tmp_path / 'test_audio'
    src.mkdir()
    for fn, a in wavs.items():
        wavfile.write(str(src / fn), 16000, a)
    dst = tmp_path / 'heng_pseudo'
    dst.mkdir()
    (dst / 'stale.txt').write_text('an existing pseudo dir is wiped')
    n_labels, n_small = export.make_pseudo_labels(fnames, p12, str(src), str(dst))
    g = k7['pseudo']
    assert g['printed'] == ['%d of %d pseudo labels were created.' % (n_labels, len(fnames)),
                            '%d of %d have low prob' % (n_small, len(fnames))]
    tree = {}
    for root, dirs, files in os.walk(str(dst)):
        tree[os.path.relpath(root, str(dst))] = sorted(files)
    assert tree == g['tree']
    for f, want in g['silence'].items():
        rate, data = wavfile.read(str(dst / '_background_noise_' / f))
        assert rate == want['rate'] and len(data) == want['n'] == 30 * 16000
        assert [int(v) for v in data[:8]] == want['head'] and sha1(data) == want['sha1']
    # copies are byte-identical to their sources
    some = [d for d in tree if d not in ('.', '_background_noise_') and tree[d]][0]
    fn = tree[some][0]
    assert (dst / some / fn).read_bytes() == (src / fn).read_bytes()
