import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devit_amd import ops
from devit_amd._lib import call, ptr, stream_ptr
dev = torch.device("cuda"); H = int(sys.argv[1]); B, N = 256, 198; D = H * 64; M = B * N
qkv = ops.rows_alloc(M, 3 * D, torch.bfloat16, dev, extra=128); qkv[:M] = torch.randn(M, 3 * D, device=dev).to(torch.bfloat16)
out = ops.rows_alloc(M, D, torch.bfloat16, dev); lse = torch.empty(B, H, N, device=dev)
dout = ops.rows_alloc(M, D, torch.bfloat16, dev); dout[:M] = torch.randn(M, D, device=dev).to(torch.bfloat16)
dqkv = ops.rows_alloc(M, 3 * D, torch.bfloat16, dev)
for _ in range(3):
    call("devit_attn_fwd", ptr(qkv), ptr(out), ptr(lse), None, B, N, H, 64, 0.125, 0, stream_ptr())
    call("devit_attn_bwd", ptr(qkv), ptr(out), ptr(dout), ptr(lse), None, None, ptr(dqkv), B, N, H, 64, 0.125, stream_ptr())
torch.cuda.synchronize()
