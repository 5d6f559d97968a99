"""Import alias: `import pedp_hip` loads the package that lives in
`6dof-pose-estimation-and-defect-projection_amd/` (a directory name Python cannot import
directly).  Submodules resolve there too: `import pedp_hip.compat`, `pedp_hip.synth`, ..."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "6dof-pose-estimation-and-defect-projection_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
