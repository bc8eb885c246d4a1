"""ChaCha20 expander behind ipcl::detail::fill_random (csrc/host/chacha20.hpp; bulk obfuscator randomness) against the
RFC 8439 known answers (section 2.3.2 block vector, section 2.4.2 key stream) -- CPU test, no GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("rnd") / "rnd_driver")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "pailliercryptolib_amd", "csrc", "host"),
                    os.path.join(ROOT, "tests", "cpp", "random_driver.cpp"), "-o", exe], check=True)
    return exe


def ask(driver, line):
    return subprocess.run([driver], input=line + "\n", capture_output=True, text=True, check=True).stdout.strip()


KEY = "000102030405060708090a0b0c0d0e0f101112131415161718191a1b1c1d1e1f"


def test_rfc8439_block_vector(driver):
    want = ("10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
            "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")
    assert ask(driver, f"block {KEY} 000000090000004a00000000 1 64") == want


def test_rfc8439_keystream_blocks_and_tail(driver):
    # section 2.4.2: key stream of counter 1 and 2 for nonce 00 00 00 00 00 00 00 4a 00 00 00 00; a request that is
    # not a multiple of the block size is a prefix of the same stream
    ks = ask(driver, f"block {KEY} 000000000000004a00000000 1 114")
    assert ks.startswith("224f51f3401bd9e12fde276fb8631ded8c131f823d2c06e27e4fcaec9ef3cf78"
                         "8a3b0aa372600a92b57974cded2b9334794cba40c63e34cdea212c4cf07d41b7")
    assert ks[128:128 + 64] == "69a6749f3f630f4122cafe28ec4dc47e26d4346d70b98c73f3e9c53ac40c5945"
    assert len(ks) == 228
    assert ask(driver, f"block {KEY} 000000000000004a00000000 1 70") == ks[:140]


def test_bulk_path_equals_block_by_block(driver):
    """requests of 256 bytes and more take the four-blocks-at-once path: same stream as single blocks"""
    nonce = "0102030405060708090a0b0c"
    whole = ask(driver, f"block {KEY} {nonce} 7 1000")
    single = "".join(ask(driver, f"block {KEY} {nonce} {7 + i} 64") for i in range(16))
    assert len(whole) == 2000 and whole == single[:2000]
