"""Host bookkeeping of the batched filter (FeatStore, uav_airvision_amd/csrc/msckf_batch.inc) against a dict model of the
reference's map_server on random message sequences -- compiled for the CPU with g++, no GPU, no HIP."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _featstore_source():
    text = open(os.path.join(ROOT, 'uav_airvision_amd', 'csrc', 'msckf_batch.inc')).read()
    a = text.index('struct FMeta {')
    b = text.index('struct BStream {')
    return text[a:b]


@pytest.fixture(scope='module')
def harness(tmp_path_factory):
    d = tmp_path_factory.mktemp('featstore')
    src = open(os.path.join(ROOT, 'tests', 'native', 'featstore_harness.cpp')).read().replace('FEATSTORE_SRC', _featstore_source())
    cpp = d / 'featstore_test.cpp'
    cpp.write_text(src)
    exe = d / 'featstore_test'
    subprocess.check_call(['g++', '-O1', '-g', '-std=c++17', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', str(cpp), '-o', str(exe)])
    return str(exe)


@pytest.mark.parametrize('seed,frames,per', [(1, 300, 60), (2, 300, 300), (3, 120, 5), (4, 400, 40)])
def test_featstore_matches_dict_model(harness, seed, frames, per):
    out = subprocess.run([harness, str(seed), str(frames), str(per)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith('OK'), out.stdout
    assert int(out.stdout.split()[1]) > 500
