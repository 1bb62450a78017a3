"""The 64-row half tiles of the wave-specialised NN GEMM's last round (csrc/gemm.hip) against whole tiles
(KWS_GEMM_NO_HALF=1): C must be bit-identical; two processes, the knob is read once.  Run on the GPU box."""
import sys, os, subprocess, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import numpy as np, torch
    from speech_recognition_amd import _lib
    lib = _lib.load(); S = _lib.stream_ptr()
    h = hashlib.sha256()
    for (M, K, N) in [(33692, 128, 128), (33856, 192, 384), (34000, 128, 192), (40000, 64, 128), (99328, 256, 256), (9216, 512, 512), (70000, 160, 64)]:
        g = torch.Generator(device='cuda'); g.manual_seed(M + K)
        A = torch.randn(M, K, device='cuda', generator=g); W = torch.randn(K, N, device='cuda', generator=g) * 0.1
        C = torch.full((M, N), float('nan'), device='cuda')
        rows = lib.kws_gemm_nn_stats_rows(M, K, N)
        part = torch.zeros(rows, 2, N, device='cuda')
        _lib.call("kws_gemm_nn_f32", _lib.ptr(A), _lib.ptr(W), _lib.ptr(C), M, K, N, _lib.ptr(part), S)
        torch.cuda.synchronize()
        assert torch.isfinite(C).all()
        h.update(C.cpu().numpy().tobytes())   # (the statistics rows are grouped by workgroup: another walk, another rounding of their sum)
    print(h.hexdigest())
else:
    a = subprocess.run([sys.executable, __file__, 'x'], capture_output=True, text=True)
    e = dict(os.environ); e['KWS_GEMM_NO_HALF'] = '1'
    b = subprocess.run([sys.executable, __file__, 'x'], capture_output=True, text=True, env=e)
    print('half :', a.stdout.strip()[-64:], a.stderr[-300:] if a.returncode else '')
    print('whole:', b.stdout.strip()[-64:], b.stderr[-300:] if b.returncode else '')
    ok = a.returncode == 0 and b.returncode == 0 and a.stdout.strip()[-64:] == b.stdout.strip()[-64:]
    print('C bit-identical:', ok)
    sys.exit(0 if ok else 1)
