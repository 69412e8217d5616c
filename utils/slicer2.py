from some_amd.utils.slicer2 import Slicer, get_rms  # noqa: F401  (drop-in shim for reference utils/slicer2.py)
