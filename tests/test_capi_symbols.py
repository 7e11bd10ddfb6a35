"""The C-ABI library loads on a machine without a GPU, exports every function that
include/uda_hip.h declares, and the ctypes structures have the C layout.  No compute calls."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from common import ROOT
from uda_amd import capi

HEADER = os.path.join(ROOT, "include", "uda_hip.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uda_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib_path():
    if not os.path.exists(capi.LIB_PATH):
        capi.build()
    return capi.LIB_PATH


def test_header_and_binding_agree():
    declared = _declared_functions()
    assert declared, "no functions parsed from the header"
    assert sorted(capi.EXPORTS) == declared


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in _declared_functions():
        assert hasattr(lib, name), "libuda_hip.so lacks %s" % name


def test_struct_layouts_match_c(tmp_path):
    csrc = tmp_path / "sz.c"
    csrc.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "uda_hip.h"\n'
                    'int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(uda_buf_desc_t), sizeof(uda_op_t),'
                    'sizeof(uda_drop_site_t), sizeof(uda_model_t), offsetof(uda_op_t, w_off), offsetof(uda_op_t, fuse_w),'
                    'offsetof(uda_model_t, arena_floats), offsetof(uda_model_t, nms_soft_sigma));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(csrc), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(capi.BufDesc), ctypes.sizeof(capi.Op), ctypes.sizeof(capi.DropSite),
            ctypes.sizeof(capi.Model), capi.Op.w_off.offset, capi.Op.fuse_w.offset,
            capi.Model.arena_floats.offset, capi.Model.nms_soft_sigma.offset]
    assert got == want


def test_create_fails_loudly_without_gpu(lib_path):
    """No CPU fallback: without a HIP device uda_create must return an error, not a handle."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import ctypes as C\n"
            "from uda_amd import capi\n"
            "lib = capi.load()\n"
            "m = capi.Model(); m.abi_version = capi.UDA_ABI_VERSION; m.num_levels = 1; m.chunk_images = 1\n"
            "m.max_images = 1; m.mc_samples = 1\n"
            "import numpy as np\n"
            "w = np.zeros(4, np.float32); h = C.c_void_p()\n"
            "rc = lib.uda_create(C.byref(m), (capi.BufDesc * 1)(), 1, (capi.Op * 1)(), 0, (capi.DropSite * 1)(),\n"
            "                    w.ctypes.data, 4, w.ctypes.data, 0, C.byref(h))\n"
            "print('rc', rc, 'handle', h.value, lib.uda_last_error(None).decode())\n" % ROOT)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "rc 1" in out.stdout and "handle None" in out.stdout, out.stdout
