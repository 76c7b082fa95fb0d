"""The driver's entry points (__graft_entry__.build / smoke) and the loader's HIP-runtime rule.

The torch wheel ships its own libamdhip64; a process that dlopens libunetpp_hip.so BEFORE importing torch ends up with
two HIP runtimes, and the one initialised second sees no device (`unetpp_create`: "no HIP device available").  The
loader therefore imports torch first (unet-_amd/_lib.py share_torch_hip_runtime)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(code, timeout=600):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_loader_imports_torch_before_the_library():
    r = _run("import sys\n"
             "from unet_amd import _lib\n"
             "assert 'torch' not in sys.modules\n"
             "_lib.load()\n"
             "assert 'torch' in sys.modules\n"
             "maps = open('/proc/self/maps').read()\n"
             "hips = sorted({l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l})\n"
             "print(hips)\n"
             "assert len(hips) == 1, hips\n")
    assert r.returncode == 0, r.stdout + r.stderr


def test_build_exports_every_abi_symbol():
    r = _run("import __graft_entry__ as g\ng.build()\n")
    assert r.returncode == 0 and "unetpp_version ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_build_then_smoke_in_one_process():
    r = _run("import __graft_entry__ as g\ng.build()\ng.smoke()\nprint('both ok')\n")
    assert r.returncode == 0 and "both ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_library_loaded_before_torch_still_finds_the_device():
    r = _run("from unet_amd import _lib\n"
             "_lib.load()\n"
             "import torch\n"
             "from unet_amd.nested_unet import NestedUNet\n"
             "from unet_amd import synthetic\n"
             "m = NestedUNet(3, max_batch=1, max_hw=(32, 32)).to('cuda:0')\n"
             "m.load_state_dict(synthetic.make_state_dict(3, 3, True, 2))\n"
             "print(tuple(m.segment(torch.rand(1, 3, 32, 32, device='cuda:0')).shape))\n")
    assert r.returncode == 0 and "(1, 32, 32)" in r.stdout, r.stdout + r.stderr
