"""The device libm (maray_amd/csrc/maray_libm.h) compiled for the host and compared bit for bit with the
system libm — the libm the reference's f64::sin/exp/ln reach (src/lib.rs:648-650)."""
import os
import subprocess

from conftest import ROOT


def build_checker(tmp):
    exe = os.path.join(tmp, 'libm_check')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-mfma', '-ffp-contract=off', '-fno-builtin',
                           '-I' + os.path.join(ROOT, 'maray_amd', 'csrc'),
                           os.path.join(ROOT, 'tests', 'native', 'libm_check.cpp'), '-o', exe, '-lm'])
    return exe


def test_sin_exp_log_match_glibc_bit_for_bit(tmp_path):
    exe = build_checker(str(tmp_path))
    out = subprocess.run([exe, '300000'], capture_output=True, text=True)
    print(out.stdout)
    assert out.returncode == 0, out.stdout[-2000:]
    assert 'sin mismatches 0, exp mismatches 0, log mismatches 0' in out.stdout


def test_step_sin_sign_next_to_every_multiple_of_half_pi(tmp_path):
    """Exhaustive: all doubles within 2 ulps of k*pi/2 for every k inside reduce_sincos's range."""
    exe = build_checker(str(tmp_path))
    out = subprocess.run([exe, '1000', 'exhaustive'], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert 'walked 335544295 doubles' in out.stdout
