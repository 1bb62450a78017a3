"""The PRODUCT's feature path end to end against the oracle: AudioProcessor batches ('mfcc', 'spec', 'mfcc_and_raw')
and AudioConverter.load() are computed by the device kernel from the tables that speech_recognition_amd/features.py
builds (the kernel-level tests feed the kernel the ORACLE's tables, so a slip in the product's own table code would
have passed them - VERDICT r1, weak 1).  Reference: input_data.py:361-381, 395-541; audio.py:15-28."""
import numpy as np
import pytest
import torch

from oracle import features as OF

pytestmark = pytest.mark.gpu


def _processor(rep, n_mel, n_keep, win_ms=30.0, stride_ms=10.0, n_bank=2048):
    import bench
    from speech_recognition_amd.input_data import AudioProcessor, prepare_words_list
    from speech_recognition_amd.model import prepare_model_settings
    dev = torch.device("cuda", 0)
    settings = prepare_model_settings(label_count=len(prepare_words_list(bench.WANTED)), sample_rate=16000,
                                      clip_duration_ms=1000, window_size_ms=win_ms, window_stride_ms=stride_ms,
                                      dct_coefficient_count=n_mel, num_log_mel_features=n_keep, output_representation=rep)
    spec = bench.build_synthetic(dev, n_bank, seed=59185)
    proc = AudioProcessor(spec, 13.0, 60.0, bench.WANTED, 10.0, 0.0, settings, output_representation=rep, device=dev)
    return proc, settings, spec


def _validation_clips(proc, spec, n):
    """What get_data(mode='validation') feeds the feature stage: bank rows of the partition in index order, silence
    entries zeroed (input_data.py:459-461, 503-504), no shift, no noise."""
    rows = proc._rows['validation'][:n]
    clips = spec['bank'].clips[torch.from_numpy(rows.astype(np.int64)).cuda()].cpu().numpy().astype(np.float64)
    clips[proc._silence['validation'][:n]] = 0.0
    return clips


@pytest.mark.parametrize("n_mel,n_keep,win_ms,stride_ms", [(80, 60, 30.0, 10.0), (40, 40, 30.0, 10.0), (40, 40, 25.0, 15.0)])
def test_audio_processor_mfcc_batches_match_oracle(n_mel, n_keep, win_ms, stride_ms):
    from speech_recognition_amd.utils import data_gen
    proc, settings, spec = _processor('mfcc', n_mel, n_keep, win_ms, stride_ms)
    gen = data_gen(proc, None, batch_size=37, mode='validation')
    X, y = next(gen)
    X = np.asarray(X)
    F = settings['spectrogram_length']
    assert F == {30.0: 98, 25.0: 66}[win_ms] and X.shape == (37, F * n_keep) == (37, settings['fingerprint_size'])
    clips = _validation_clips(proc, spec, 37)
    tables = OF.tables_path_b(settings['window_size_samples'], n_mel, n_keep)
    ref = OF.features(clips, tables, settings['window_stride_samples'], dtype=np.float64).reshape(37, -1)
    assert np.abs(X - ref).max() < 2e-3             # DCT outputs are O(10): same bar as the kernel-level test
    lab = np.asarray(y).argmax(1)
    assert np.array_equal(lab, proc._labels['validation'][:37])
    # second batch continues where the first stopped (utils.py:38-40)
    X2, _ = next(gen)
    ref2 = OF.features(_validation_clips(proc, spec, 74)[37:], tables, settings['window_stride_samples'], dtype=np.float64)
    assert np.abs(np.asarray(X2) - ref2.reshape(37, -1)).max() < 2e-3


def test_audio_processor_spec_and_mfcc_and_raw_outputs_match_oracle():
    from speech_recognition_amd.utils import data_gen
    proc, settings, spec = _processor('spec', 80, 60)
    X, _ = next(data_gen(proc, None, batch_size=9, mode='validation'))
    clips = _validation_clips(proc, spec, 9)
    tables = OF.tables_path_b(480, 80, 60)
    mag = OF.stft_magnitude(clips, tables, 160, np.float64).reshape(9, -1)
    X = np.asarray(X)
    assert X.shape == (9, 98 * 257)
    assert np.abs(X - mag).max() < 2e-5 * max(1.0, np.abs(mag).max())
    proc2, settings2, spec2 = _processor('mfcc_and_raw', 80, 60)
    (Xm, Xr), _ = next(data_gen(proc2, None, batch_size=9, mode='validation'))
    clips2 = _validation_clips(proc2, spec2, 9)
    assert np.array_equal(np.asarray(Xr), clips2.astype(np.float32))                 # raw arm: the clip itself, bit for bit
    ref = OF.features(clips2, tables, 160, dtype=np.float64).reshape(9, -1)
    assert np.abs(np.asarray(Xm) - ref).max() < 2e-3


def test_audio_processor_training_batch_features_follow_the_augmented_clip():
    """Training mode: the features are those of the AUGMENTED clip (the 'raw' arm of the same batch is the witness)."""
    from speech_recognition_amd.utils import data_gen
    proc, settings, spec = _processor('mfcc_and_raw', 80, 60)
    np.random.seed(77)
    (Xm, Xr), _ = next(data_gen(proc, None, batch_size=21, mode='training', pseudo_frequency=0.6))
    raw = np.asarray(Xr).astype(np.float64)
    ref = OF.features(raw, OF.tables_path_b(480, 80, 60), 160, dtype=np.float64).reshape(21, -1)
    assert np.abs(np.asarray(Xm) - ref).max() < 2e-3
    assert np.abs(raw).max() > 0


def test_audio_converter_load_matches_oracle_path_a(tmp_path):
    """audio.py:15-28: decode_wav -> AudioSpectrogram(480, 160, squared) -> Mfcc(40): [1, 98, 40]."""
    from scipy.io import wavfile
    from speech_recognition_amd.audio import AudioConverter
    rng = np.random.RandomState(4)
    tables = OF.tables_path_a(480, 16000, 40, 40)
    conv = AudioConverter()
    for name, n in (("full", 16000), ("short", 11000), ("long", 19000)):
        x = 0.3 * np.sin(2 * np.pi * 440.0 * np.arange(n) / 16000.0)
        x = x + 0.05 * rng.randn(len(x))
        pcm = np.clip(x * 32768.0, -32768, 32767).astype(np.int16)
        fn = str(tmp_path / (name + ".wav"))
        wavfile.write(fn, 16000, pcm)
        got = conv.load(fn)
        assert got.shape == (1, 98, 40) and got.dtype == np.float32
        clip = np.zeros(16000)
        m = min(len(pcm), 16000)
        clip[:m] = pcm[:m].astype(np.float64) / 32768.0                               # DecodeWav, desired_samples=16000
        ref = OF.features(clip[None], tables, 160, dtype=np.float64)
        assert np.abs(got - ref).max() < 2e-3, name
    # batched entry: the same rows through convert()
    xb = torch.from_numpy((0.1 * rng.randn(5, 16000)).astype(np.float32)).cuda()
    out = conv.convert(xb).cpu().numpy()
    ref = OF.features(xb.cpu().numpy().astype(np.float64), tables, 160, dtype=np.float64)
    assert np.abs(out - ref).max() < 2e-3


def test_whole_partition_larger_than_one_launch():
    """get_data(how_many=-1) on a partition of more than 65535 entries (the augment grid's y limit): several
    launches, every row still the right clip."""
    import bench
    from speech_recognition_amd.input_data import AudioProcessor, ClipBank, prepare_words_list
    from speech_recognition_amd.model import prepare_model_settings
    dev = torch.device("cuda", 0)
    L = 1000                                                        # short clips keep this at 0.5 GB
    n = 70000
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    bank = torch.randn((512, L), generator=g, device=dev)
    settings = prepare_model_settings(label_count=len(prepare_words_list(bench.WANTED)), sample_rate=16000,
                                      clip_duration_ms=1000 * L / 16000.0, window_size_ms=30.0, window_stride_ms=10.0,
                                      dct_coefficient_count=40, num_log_mel_features=40, output_representation='raw')
    assert settings['desired_samples'] == L
    rows = np.random.RandomState(0).randint(0, 512, n)
    index = {'training': [(0, 'yes')], 'pseudo': [], 'testing': [],
             'validation': [(int(r), bench.WANTED[int(r) % 10]) for r in rows]}
    proc = AudioProcessor({'bank': ClipBank(bank, [], dev), 'index': index}, 0.0, 0.0, bench.WANTED, 10.0, 0.0, settings,
                          output_representation='raw', device=dev)
    X, y = proc.get_data(-1, 0, 0.0, 0.0, 0.0, 0.0, 0.0, [0, 0], 'validation', None)
    Xt = X.wait()
    torch.cuda.synchronize()
    assert tuple(Xt.shape) == (n, L)
    assert torch.equal(Xt, bank[torch.from_numpy(rows).to(dev)])


def test_processor_closed_and_replaced_keeps_generating():
    """Round 5: a second AudioProcessor built after the first was closed (val-acc parity's seeds, a driver that re-creates its
    generator) failed in its first parameter upload with 'operation not permitted on an event last recorded in a capturing
    stream': torch's pinned-memory cache still held events recorded on the closed processor's stream, whose handle the new
    stream had taken.  close() now lets the cache retire them first; a closed processor refuses get_data cleanly."""
    from speech_recognition_amd import _lib
    from speech_recognition_amd.utils import data_gen
    first = None
    for round_ in range(4):
        proc, settings, spec = _processor('raw', 80, 60, n_bank=1024)
        gen = data_gen(proc, None, batch_size=64, mode='training', pseudo_frequency=0.6)
        for _ in range(3):
            X, y = next(gen)
        X = np.asarray(X)
        assert X.shape == (64, 16000) and np.isfinite(X).all() and np.abs(X).max() > 0
        if first is None:
            first = X.shape
        proc.close()
        with pytest.raises(_lib.KwsError):
            proc.get_data(4, 0, 0.0, 0.0, 0.0, 0.0, 0.0, [0, 0], 'validation', None)
        proc.close()                                     # idempotent
