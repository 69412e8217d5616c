"""Drop-in shim for ``from utils import build_object_from_class_name`` (reference utils/__init__.py:221-230)."""
from some_amd.utils import build_object_from_class_name  # noqa: F401
