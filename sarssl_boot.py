"""Makes the product package importable as ``sar_ssl_amd``.

The package directory is ``sar-ssl_amd/`` (the name the layout contract fixes), which is not a
valid Python identifier.  A ``sar_ssl_amd -> sar-ssl_amd`` symlink is committed for
convenience; if a checkout/snapshot drops the symlink this module registers the package from
the hyphenated directory instead.  Import this before ``import sar_ssl_amd``.
"""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

if "sar_ssl_amd" not in sys.modules and not os.path.isdir(os.path.join(_ROOT, "sar_ssl_amd")):
    _dir = os.path.join(_ROOT, "sar-ssl_amd")
    _spec = importlib.util.spec_from_file_location(
        "sar_ssl_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules["sar_ssl_amd"] = _mod
    _spec.loader.exec_module(_mod)
