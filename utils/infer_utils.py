from some_amd.utils.infer_utils import *  # noqa: F401,F403  (drop-in shim for reference utils/infer_utils.py)
