"""Registers the phnet_amd drop-in modules under the reference's import paths, so that
`from libs.models.Router4OL import RouterOL` (`libs.models.Router4OLV2` for testOLV3.py:11), `from libs.utils.loss4OLV3 import Criterion4OL` and
`from libs.ops import nms` (trainOL.py:12-14, testOL.py:19-23, Router4OL.py:10) resolve to the HIP-backed classes."""
import importlib
import sys

_MAP = {
    "libs.models.Router4OL": "phnet_amd.libs.models.Router4OL",
    "libs.models.Router4OLV2": "phnet_amd.libs.models.Router4OLV2",
    "libs.models.fpnV2": "phnet_amd.libs.models.fpnV2",
    "libs.models.resnet": "phnet_amd.libs.models.resnet",
    "libs.models.fpn": "phnet_amd.libs.models.fpn",
    "libs.models.Router": "phnet_amd.libs.models.Router",
    "libs.models.utils.dynamic_head": "phnet_amd.libs.models.utils.dynamic_head",
    "libs.models.utils.transformer": "phnet_amd.libs.models.utils.transformer",
    "libs.utils.loss4OLV3": "phnet_amd.libs.utils.loss4OLV3",
    "libs.utils.loss4OL": "phnet_amd.libs.utils.loss4OL",
    "libs.utils.loss4OLV2": "phnet_amd.libs.utils.loss4OLV2",
    "libs.utils.lane": "phnet_amd.libs.utils.lane",
    "libs.ops": "phnet_amd.libs.ops",
    "libs.ops.nms": "phnet_amd.libs.ops.nms",
}


def install(force: bool = True):
    """Call BEFORE the caller script imports `libs...`.  With force=False existing entries are kept."""
    done = []
    for alias, target in _MAP.items():
        if not force and alias in sys.modules:
            continue
        sys.modules[alias] = importlib.import_module(target)
        done.append(alias)
    return done
