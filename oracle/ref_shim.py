"""Import shim for the REAL reference (test infrastructure only; this container only).

Nothing in the product path (``sar-ssl_amd/``) may import this module.  It exists so the
oracle restatement (``oracle/sarssl_oracle.py``) can be validated against the reference's
own Python code and so golden vectors can be generated (``oracle/make_golden.py``).
The reference tree (``/root/reference``) does not exist on the GPU box; ``available()``
tells callers whether it can be used.

The reference does not import as shipped: ``code/model.py:12-15`` imports four modules that
are not in the tree and several third-party packages are absent here.  They are all unused on
the pretraining path, so empty stand-in *modules* (never stand-in source files) are
registered in ``sys.modules`` before the import.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("SARSSL_REFERENCE", "/root/reference")
REF_CODE = os.path.join(REF_ROOT, "code")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_CODE, "model.py"))


def _stub(name, **attrs):
    if name in sys.modules:
        return
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m


_loaded = None


def load():
    """Returns (model, learner, utils_module) modules of the reference."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    import torch

    if REF_CODE not in sys.path:
        sys.path.insert(0, REF_CODE)
    _stub("timm"); _stub("timm.models")
    _stub("timm.models.layers", trunc_normal_=torch.nn.init.trunc_normal_)
    _stub("common.NBC", NBC=None); _stub("common.FNSSL", FNblock=None); _stub("common.UNet", UNet=None)
    _stub("common.CNN", resnet50=None, res2net50=None, densenet121=None)
    _stub("torchaudio"); _stub("soundfile")
    _stub("torchmetrics"); _stub("torchmetrics.functional"); _stub("torchmetrics.functional.audio")
    _stub("torchmetrics.functional.audio.pesq", perceptual_evaluation_speech_quality=None)
    import model as ref_model            # noqa: E402
    import learner as ref_learner        # noqa: E402
    import common.utils_module as ref_um  # noqa: E402
    _loaded = (ref_model, ref_learner, ref_um)
    return _loaded
