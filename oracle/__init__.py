"""CPU oracle for the DASR SRN training-step hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain fp32 PyTorch-CPU restatement
of the reference algorithm (ShuhangGu/DASR, codes/SRN).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker -- never as the thing measured or shipped.  The
product path (``dasr_amd``) never imports ``oracle``.

Pinning status ("Oracle"):
  * RRDBNet, NLayerDiscriminator, GaussianFilter/FilterLow/FilterHigh, GANLoss,
    SRModel step, DASR_Model step (gaussian + wavelet fs, random-init VGG19-54):
    PINNED against the reference itself, imported in the build container with
    sys.modules stubs (``oracle/ref_import.py``); fixtures in ``tests/golden``
    were produced by ``oracle/gen_golden.py`` running the *reference* code.
  * Haar DWT sub-band order/sign (third-party ``pytorch_wavelets``, un-vendored,
    un-pinned in the reference) and pretrained VGG19 / LPIPS weights:
    PARITY UNPINNED.  The Haar convention used is documented in
    ``oracle/nets.py::HaarDWT``.
"""
