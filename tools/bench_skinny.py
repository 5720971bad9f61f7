"""Kernel time of the decoders' 100-row f32 GEMMs: gemm_f32_skinny_kernel against gemm_f32_kernel<64,64> (run under
`rocprofv3 --kernel-trace --stats`: a Python launch loop is host-bound at these sizes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from openvis_amd import ops

lib = ops._lib.lib()
for (M, N, K) in [(100, 256, 256), (100, 768, 256), (100, 2048, 256), (100, 256, 2048)]:
    a, w, b = torch.randn(M, K).cuda(), torch.randn(N, K).cuda(), torch.randn(N).cuda()
    for mode in (1, 2, 3, 4):            # 1: 32-row workgroups (K <= 512) | 2: 128-row form | 3: gemm_f32_kernel<64,64> for K <= 512
        lib.ovis_set_skinny_gemm(mode)
        for _ in range(20):
            ops.gemm_nt(a, w, b)
        torch.cuda.synchronize()
lib.ovis_set_skinny_gemm(1)
