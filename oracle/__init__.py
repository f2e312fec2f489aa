"""CPU oracle for the pySDR receiver hot path.  TEST INFRASTRUCTURE ONLY.

Nothing in ``pysdr_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it,
and only as the checker / the CPU number reported beside the GPU number.

PARITY STATUS: **partially pinned**.  The arithmetic of the reference's hot path
lives in module ``sig_proc`` of the un-vendored, un-pinned repo ``aa2il/libs``
(``/root/reference/receiver.py:39,45``; ``README.md:47-49``), which is absent
from ``/root/reference``.  This oracle is therefore a restatement of the DSP spec
written down in ``DESIGN.md`` section 3, anchored on the only in-tree pins:

  P1 ``srates.py:35-74``        39-row known-answer table for ``up_dn``
  P2 ``sigs/nfm.m:124-127``     NFM discriminator formula
  P3 ``sigs/agc.m:6-12``        AGC loop filter ``filter(beta,[1 beta-1],x)``, beta=.1
  P4 ``sigs/iir.py:83-125``     chunked == one-shot property for stateful stages
  P5 ``rtty.py:839-841``        windowed FFT -> ``10*log10(re^2+im^2)`` after fftshift
  P6 in-tree integer/float arithmetic of ``params.py:405-468``,
     ``utils.py:277-289``, ``receiver.py:200,250-252``, ``Plotting.py:370-375,
     539-626,689-695``

plus ``scipy.signal.upfirdn`` as an independent implementation of the rational
resampler.  Everything else (mixer sign, FIR design, demod formulae, AGC
constants, PSD window/normalisation) is "parity unpinned": it is parity with
this oracle, not with ``sig_proc``.
"""
