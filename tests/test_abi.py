"""The C-ABI library loads and exports every symbol include/*.h declares; the
render entry points fail loudly without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

import maray_amd as M
from conftest import ROOT
from marayb import encode, x


def declared_functions():
    names = []
    for hdr in ('maray_hip.h',):
        src = open(os.path.join(ROOT, 'include', hdr)).read()
        src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
        for m in re.finditer(r'^\s*(?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\**\s*(maray_[a-z0-9_]+)\s*\(', src, flags=re.M):
            names.append(m.group(1))
    return sorted(set(names))


def test_every_declared_symbol_is_exported():
    names = declared_functions()
    assert len(names) >= 29
    L = C.CDLL(M.lib_path())
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_product_does_not_link_or_load_the_oracle():
    import subprocess
    out = subprocess.run(['ldd', M.lib_path()], capture_output=True, text=True).stdout
    assert 'oracle' not in out
    for root, _, files in os.walk(os.path.join(ROOT, 'maray_amd')):
        for f in files:
            if f.endswith(('.cpp', '.hpp', '.h', '.hip', '.py')):
                assert 'oracle' not in open(os.path.join(root, f), errors='ignore').read().lower(), f


@pytest.mark.skipif(M.device_count() > 0, reason='a GPU is present')
def test_render_fails_loudly_without_a_device():
    t = M.Scene(encode((8, 8), [x(), x(), x()])).lower()
    for backend in (M.BACKEND_TAPE, M.BACKEND_TAPE_SMEM):
        with pytest.raises(M.MarayError) as e:
            M.Context(t, backend=backend)
        assert e.value.code == -8
    with pytest.raises(M.MarayError):
        M.gen_to_image(M.Scene(encode((8, 8), [x(), x(), x()])))
