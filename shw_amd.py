"""Import alias: the package directory name required by the build contains hyphens, which the
`import` statement cannot spell.  `import shw_amd` gives the same module object."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
_pkg = importlib.import_module("sphere-homeomorphic-wasserstein-distance-for-point-cloud-registration_amd")
sys.modules[__name__] = _pkg
