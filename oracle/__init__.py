"""TEST INFRASTRUCTURE ONLY.

CPU (PyTorch fp32) restatement of the DeViT hot path, used as the parity
checker by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py``.  Nothing under ``devit_amd/`` may import this package: the
product path runs on the HIP library only and fails loudly without it.

Parity status: PINNED by golden vectors in ``tests/golden/*.npz`` that were
produced by importing the reference's own modules (``/root/reference``,
``models/de_vit.py``, ``utils/losses.py``) in the build container with
``tests/golden/make_golden.py``.  The reference ships no tests or golden
vectors of its own, so "outputs of the reference itself run
here" is the pin.
"""
