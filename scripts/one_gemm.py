import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_recognition_amd import _lib
lib = _lib.load()
M, K, N = [int(v) for v in sys.argv[1:4]]
S = _lib.stream_ptr()
A = torch.randn(M, K, device='cuda'); W = torch.randn(K, N, device='cuda') * 0.1; C = torch.empty(M, N, device='cuda')
G = torch.randn(M, N, device='cuda'); dW = torch.empty(K, N, device='cuda')
part = torch.empty(lib.kws_gemm_num_row_tiles(M) * 2 * N, device='cuda')
ws = torch.empty(int(lib.kws_gemm_tn_workspace_floats(M, K, N)), device='cuda')
for _ in range(3):
    _lib.call("kws_gemm_nn_f32", _lib.ptr(A), _lib.ptr(W), _lib.ptr(C), M, K, N, _lib.ptr(part), S)
    _lib.call("kws_gemm_tn_f32", _lib.ptr(A), _lib.ptr(G), _lib.ptr(dW), M, K, N, _lib.ptr(ws), S)
torch.cuda.synchronize()
