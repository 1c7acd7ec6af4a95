"""Import shim: the package directory is `video-filler_amd/` (not a valid Python identifier), so this module
presents it as the package `video_filler_amd`."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "video-filler_amd")]
__package__ = __name__
with open(_os.path.join(__path__[0], "__init__.py")) as _fh:
    exec(compile(_fh.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
