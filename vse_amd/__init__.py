"""Import alias for the product package.

The product sources live in `video-subtitle-extractor_amd/` (the directory name the build contract asks
for; a hyphen is not a legal Python identifier), so `import vse_amd` maps its submodule search path there:
`vse_amd.engine` is `video-subtitle-extractor_amd/engine.py`, etc.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "video-subtitle-extractor_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
