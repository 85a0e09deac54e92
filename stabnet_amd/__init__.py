"""Import alias for the package directory `deep-online-video-stabilization_amd/`.

The directory name required by the project layout contains hyphens and so cannot be
named in an `import` statement; this stub makes it importable as `stabnet_amd`
(`stabnet_amd.<module>` resolves inside that directory).
"""
import os as _os

_here = _os.path.dirname(_os.path.abspath(__file__))
_real = _os.path.join(_os.path.dirname(_here), "deep-online-video-stabilization_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
