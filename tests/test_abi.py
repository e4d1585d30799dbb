"""The C-ABI library loads on a CPU-only box and exports every symbol include/airvision.h declares
(no compute calls here: there is no GPU)."""
import ctypes
import os
import re

from conftest import ROOT


def _header_functions():
    src = open(os.path.join(ROOT, 'include', 'airvision.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    names = re.findall(r'\b(av_[a-z0-9_]+)\s*\(', src)
    return sorted(set(names))


def test_header_declares_the_expected_surface():
    names = _header_functions()
    for must in ('av_pyramid_build', 'av_lk_track', 'av_fast_detect', 'av_undistort_points', 'av_distort_points',
                 'av_frontend_create', 'av_frontend_step', 'av_frontend_step_host', 'av_frontend_push_imu',
                 'av_frontend_read_features', 'av_last_error'):
        assert must in names


def test_library_exports_every_declared_symbol():
    from uav_airvision_amd import _native
    assert os.path.exists(_native.LIB_PATH), 'run python -m uav_airvision_amd.build'
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in _header_functions():
        assert hasattr(lib, name), 'missing export: ' + name


def test_python_binding_covers_the_header():
    from uav_airvision_amd import _native
    assert sorted(_native.SIGNATURES) == _header_functions()
    _native.lib()          # resolves every symbol and sets prototypes


def test_struct_layouts_match_the_header():
    """sizeof the ctypes mirrors == what the C compiler lays out (checked by compiling a probe)."""
    import subprocess
    import tempfile
    from uav_airvision_amd import _native
    probe = '#include "airvision.h"\n#include <stdio.h>\nint main(){printf("%zu %zu\\n", sizeof(av_pyr_layout), sizeof(av_frontend_config));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, 'p.c'); exe = os.path.join(d, 'p')
        open(c, 'w').write(probe)
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe])
        a, b = [int(v) for v in subprocess.check_output([exe]).split()]
    assert ctypes.sizeof(_native.PyrLayout) == a
    assert ctypes.sizeof(_native.FrontendConfig) == b


def test_host_only_entry_points_work_without_a_gpu():
    from uav_airvision_amd import _native as N
    lay = N.PyrLayout()
    assert N.lib().av_pyramid_layout(752, 480, 4, ctypes.byref(lay)) == 0
    assert list(lay.w[:4]) == [752, 376, 188, 94] and list(lay.h[:4]) == [480, 240, 120, 60]
    assert list(lay.pitch[:4]) == [784, 416, 224, 128]
    assert lay.bytes % 256 == 0 and lay.bytes >= 784 * 512 + 416 * 272 + 224 * 152 + 128 * 92
    # errors are codes + text, never exceptions across the boundary
    assert N.lib().av_pyramid_layout(20, 20, 4, ctypes.byref(lay)) == N.AV_E_INVALID
    assert b'too small' in N.lib().av_last_error()
    assert N.lib().av_pyramid_layout(752, 480, 9, ctypes.byref(lay)) == N.AV_E_INVALID
    assert b'gfx950' in N.lib().av_version()


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'uav_airvision_amd')
    for base, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(base, f), errors='replace').read()
                assert 'import oracle' not in text and 'from oracle' not in text and 'oracle/' not in text.replace('the CPU oracle', ''), f
