"""Pins the oracle to the reference's own committed known-answer data (tests/golden/*.data are byte copies of
/root/reference/test/bootstrap_modules/*.data; the expected integers are its plain*.txt / sum.txt / diff.txt /
carry.txt).  The LWE key is regenerated from the reference's hard-coded seed {100, 20032, 21341}
(src/bootstrap_modules.cpp:52-55) by oracle/gen_fixture_key.cpp; SURVEY.md section 8(c), appendix B."""
import os
import subprocess

import numpy as np
import pytest

from conftest import full_adder

EXPECTED = dict(cloud1=9876, cloud2=686, cloud3=1287, cloud4=2000, allOne=0xFFFFFFFF, allZero=0, lsbOne=1,
                lsbZero=0xFFFFFFFE, sum=10562, diff=9190, carry=3448)
FRESH = ("cloud1", "cloud2", "cloud3", "cloud4", "allOne", "allZero", "lsbOne", "lsbZero")


def test_fixture_key_regenerates(O):
    exe = os.path.join(O.ORACLE_DIR, "gen_fixture_key")
    if not os.path.exists(exe):
        pytest.skip("g++ build of gen_fixture_key not available")
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.strip()
    assert out == "".join(str(int(b)) for b in O.fixture_key())
    assert len(out) == 630


@pytest.mark.parametrize("name", sorted(EXPECTED))
def test_fixture_format_and_decrypt(O, name):
    # record = int32 type 42 | int32 a[630] | int32 b | double variance ; index 0 = MSB (src/bootstrap_modules.cpp:95)
    types, words, var = O.load_fixture_records(name + ".data")
    assert np.all(types == 42)
    key = O.fixture_key()
    ph = np.array([O.lib().oracle_lwe_phase(O.p32(key), 630, O.p32(np.ascontiguousarray(r))) for r in words], np.int32)
    assert O.bits_to_int_msb_first(ph > 0) == EXPECTED[name]
    noise = np.abs(np.abs(ph / 2.0**32) - 0.125).max()
    if name in FRESH:
        assert noise < 2e-4 and np.allclose(var, 9.314704e-10, rtol=1e-6)   # alpha = 3.052e-5
    else:
        assert noise < 0.0086                                                # bootstrapped outputs of the reference


def test_text_fixtures_agree(O):
    G = O.GOLDEN
    assert int(open(os.path.join(G, "plain1.txt")).read()) == 9876
    assert int(open(os.path.join(G, "plain2.txt")).read()) == 686
    assert int(open(os.path.join(G, "plain3.txt")).read()) == 1287
    assert int(open(os.path.join(G, "plain4.txt")).read()) == 2000
    assert int(open(os.path.join(G, "sum.txt")).read().strip(), 2) == 10562
    assert int(open(os.path.join(G, "diff.txt")).read().strip(), 2) == 9190
    assert int(open(os.path.join(G, "carry.txt")).read().strip(), 2) == 3448


def test_oracle_adder_and_subtractor_on_reference_ciphertexts(O, sk128):
    """The reference's FullAdder / subtractor (src/bootstrap_modules.cpp:20-44, 412-482) evaluated by the oracle on
    the reference's own input ciphertexts, with OUR bootstrapping key for the reference's LWE key, must decrypt to
    the reference's sum / carry / diff and stay inside the reference's post-bootstrap noise envelope.
    CPU budget: the oracle takes ~0.25 s per gate, so this test runs the low 16 bits of the 32-bit words (all
    three fixture values are < 2^14, so the low halves carry the whole answer); the full 32-bit circuits run
    on the GPU in tests/test_gpu_parity.py and are compared bit-for-bit with this oracle."""
    from conftest import threaded_multi
    p, K, orc = sk128
    lo = slice(16, 32)   # MSB-first arrays: indices 16..31 are the low 16 bits
    _, c1, _ = O.load_fixture_records("cloud1.data")
    _, c2, _ = O.load_fixture_records("cloud2.data")
    _, all_one, _ = O.load_fixture_records("allOne.data")
    _, lsb_one, _ = O.load_fixture_records("lsbOne.data")
    c1, c2, all_one, lsb_one = c1[lo], c2[lo], all_one[lo], lsb_one[lo]
    zero = K.encrypt_bits([0], 2.0**-15, 999)[0]
    multi = threaded_multi(lambda op, x, y: orc.gates(op, x, y))
    # the adder on (c1, c2) and the two's complement of c2 are independent: evaluate them side by side
    ones = orc.gates(O.XOR, all_one, c2)                       # onesComp, src/bootstrap_modules.cpp:13-18
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(2) as ex:
        fa = ex.submit(full_adder, multi, c1, c2, zero)
        fb = ex.submit(full_adder, multi, ones, lsb_one, zero)
        (s, c), (twos, _) = fa.result(), fb.result()
    assert O.bits_to_int_msb_first(K.decrypt_bits(s)) == 10562 & 0xFFFF
    assert O.bits_to_int_msb_first(K.decrypt_bits(c)) == 3448 & 0xFFFF
    d, _ = full_adder(multi, c1, twos, zero)
    assert O.bits_to_int_msb_first(K.decrypt_bits(d)) == 9190 & 0xFFFF
    for arr in (s, d):
        assert np.abs(np.abs(K.phases(arr) / 2.0**32) - 0.125).max() < 0.03
