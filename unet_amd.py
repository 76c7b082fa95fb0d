"""Import alias: the package directory is named ``unet-_amd`` (not a valid Python identifier), so this
module makes ``import unet_amd`` / ``import unet_amd.nested_unet`` resolve into that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "unet-_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
