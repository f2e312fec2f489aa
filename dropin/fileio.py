"""``import fileio`` (``receiver.py:41``, ``pySDR.py:73``) resolves to pysdr_amd.fileio."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pysdr_amd.fileio import SDR_FILEIO, open_replay, open_writers, sdr_fileio  # noqa: E402,F401
