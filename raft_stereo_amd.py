"""Import shim: makes the hyphenated directory `raft-stereo_amd/` importable as the package
`raft_stereo_amd` (a module that defines __path__ is a package)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "raft-stereo_amd")]

from raft_stereo_amd.api import *  # noqa: E402,F401,F403
from raft_stereo_amd.api import __all__  # noqa: E402,F401
