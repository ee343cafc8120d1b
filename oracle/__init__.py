"""CPU oracle for the DDIM reconstruction hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the algorithm of the reference path named in
BASELINE.json (CLIPCondUNet forward + NoiseScheduler/DDIMSampler update, plus the
host-side codec helpers either side of it).  It is the checker for the HIP path:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  Nothing under ``clip-neural-image-conpression_amd/`` imports
it, and the product path raises when the HIP library is missing instead of
falling back to this code.

Parity status: **pinned**.  The reference is pure Python/PyTorch and imports in
the build container; ``tests/golden/make_golden.py`` ran it there (CPU, fp32) and
committed its outputs under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks every function here against those vectors.

Arithmetic is floating point (fp32), so the restatement is written with
torch-CPU ops (``torch.nn.functional``) -- the same ATen kernels the reference's
``device='cpu'`` path runs -- rather than C or numpy; the integer/byte pieces
(timestep table, .clp container, uint8 PSNR) are numpy.
"""
