"""The kernel-form policy of the host runtime (pailliercryptolib_amd/csrc/policy.cpp) on the CPU: pure host logic --
sizes x busy lanes -> kernel form, LDS claim, window -- compiled with g++ from policy.cpp alone and run here.  What it
steers: PrivateKey::decryptCRT (ipcl/pri_key.cpp:114-146), PublicKey::encrypt (pub_key.cpp:99-129), the CipherText
operators (ciphertext.cpp:135-162)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pailliercryptolib_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_kernel_form_policy(tmp_path):
    exe = str(tmp_path / "policy_tests")
    env = {k: v for k, v in os.environ.items() if not k.startswith("PGPU_")}      # the defaults, not a caller's knobs
    subprocess.run(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-DPGPU_WITH_4096=0",
                    "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                    os.path.join(ROOT, "tests", "cpp", "policy_tests.cpp"), os.path.join(CSRC, "policy.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, env=env)
    print(r.stdout)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " 0 failed" in r.stdout
