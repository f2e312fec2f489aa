"""Put this directory on PYTHONPATH (in place of aa2il/libs) and pySDR's
``from sig_proc import up_dn`` / ``import sig_proc as dsp`` (receiver.py:39,45) resolve to
the MI355X implementation."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pysdr_amd.sig_proc import (Receiver, bpf, convolver, ring_buffer2, ring_buffer3,  # noqa: E402,F401
                                signal_generator, spectrum, up_dn)
