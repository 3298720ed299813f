"""CPU oracle for the GsplatLoc hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product path (``gsplatloc_amd``)
never imports this package and fails loudly when its HIP library is missing.

PARITY UNPINNED: the arithmetic of this path lives in the third-party package
gsplat "1.3.0" (SupaVision fork of nerfstudio-project/gsplat, un-vendored
submodule ``thirdparty/gsplat``; see /root/reference/.gitmodules:4-6), whose
sources are absent from /root/reference, and the reference's own tests hold no
golden vector for rendering, loss or gradients.  The
restatement below follows the published gsplat 1.3.0 algorithm and the
reference's call sites; it is pinned by float64 finite differences, by an
independent sequential per-pixel restatement (``oracle/sequential.py``), by a
second implementation in C with hand-derived gradients
(``oracle/csrc/gsplat_oracle.c``, checked against this package's autograd) and
by invariants -- not by outputs of the reference.
"""
