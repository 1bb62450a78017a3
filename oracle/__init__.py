"""CPU oracle for the hot path of see--/speech_recognition.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the CPU number reported
beside the GPU number.  The product path (``speech_recognition_amd``) never
imports this package and fails loudly when ``libkws_hip.so`` is missing.

What it is: a NumPy restatement of the algorithms the reference runs on its
hot path.  The arithmetic of that path is
NOT in the reference repository: it lives in tensorflow-gpu==1.4.0 and
Keras==2.1.2 (reference README.md:46-47), neither of which is installed or
installable here.  Each function therefore restates the published semantics
of the TF-1.4 / Keras-2.1.2 op the reference calls and cites the reference
call site (file:line under /root/reference) it follows.

Pinning status (see DESIGN.md "Oracle"):
  * control logic (sampler draw order, data_gen epoch logic, SHA-1 split,
    index construction, settings arithmetic, label maps): PINNED by golden
    vectors captured from the reference's own input_data.py / utils.py /
    classes.py / model.prepare_model_settings imported in the build container
    (tests/golden/make_golden.py, fixtures K5).
  * architecture, layer shapes and every numeric constant (BN eps/momentum,
    RMSprop rho/eps, dropout keep-prob, L2, label smoothing, STFT frame /
    step / fft length, mel bin count, log offset, DCT scale): PINNED by the
    graph_defs embedded in the reference's shipped TensorBoard event files
    (fixtures K1), the LR schedule by the logged lr series (K3).
  * floating-point kernels (STFT, mel, DCT, conv, BN, softmax, RMSprop):
    PARITY UNPINNED by reference outputs - the reference holds no tests, no
    golden tensors and TF cannot run here.  They are cross-checked against
    independent implementations (scipy.fft, torch CPU autograd) in tests/.
  * "next" rows of SURVEY 8f: the time stretch (stretch.py: librosa's published
    algorithm, librosa itself not pinned by the reference nor installed) and the
    four further model families in net.py (SteffeNet, Conv1dResidualNet,
    MfccAndRawNet; conv_1d_spectrogram = LogMfccNet at 257 features) are PARITY
    UNPINNED the same way and cross-checked against scipy / torch autograd; the
    export tools (product module speech_recognition_amd/export.py) are pinned by
    outputs of the reference's own scripts (fixtures K7) and need no oracle.
"""
