"""Importable alias of the `pytorch_nested-unet_amd` package (hyphenated directory)."""
import importlib
import sys

_pkg = importlib.import_module("pytorch_nested-unet_amd")
sys.modules[__name__] = _pkg
