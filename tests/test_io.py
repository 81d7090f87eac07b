"""thfhe.io reads the reference's committed ciphertext files and writes them back byte for byte."""
import os

import numpy as np


def test_roundtrip_of_reference_fixture(tmp_path, O):
    from thfhe import io
    src = os.path.join(O.GOLDEN, "cloud1.data")
    recs, var = io.read_ciphertexts(src, 630)
    assert recs.shape == (32, 631) and np.allclose(var, 9.314704e-10, rtol=1e-6)
    _, words, _ = O.load_fixture_records("cloud1.data")
    assert np.array_equal(recs, words)
    dst = tmp_path / "copy.data"
    io.write_ciphertexts(dst, recs, var)
    assert open(src, "rb").read() == open(dst, "rb").read()
    key = O.fixture_key()
    ph = (recs[:, -1].astype(np.int64) - (recs[:, :-1].astype(np.int64) * key).sum(axis=1)) % 2**32
    bits = ph < 2**31          # phase > 0
    assert O.bits_to_int_msb_first(bits) == 9876


def test_seeded_lwe_key_regeneration_matches_the_reference_fixture_key():
    # thfhe.keygen.lwe_key_from_seed restates std::seed_seq + std::default_random_engine + uniform_int_distribution<int32_t>(0,1) as
    # libstdc++ implements them: with the reference's seed {100, 20032, 21341} (src/bootstrap_modules.cpp:52-55) it must reproduce the key
    # that oracle/gen_fixture_key.cpp (the C++ original) printed -- the key under which the reference's committed fixtures decrypt
    import os
    import numpy as np
    from thfhe import io, keygen
    here = os.path.dirname(os.path.abspath(__file__))
    ref = np.array([int(c) for c in open(os.path.join(here, "golden", "fixture_lwe_key.txt")).read().strip()], np.int32)
    key = keygen.lwe_key_from_seed([100, 20032, 21341], 630)
    assert np.array_equal(key, ref)
    recs, _ = io.read_ciphertexts(os.path.join(here, "golden", "cloud1.data"), 630)
    phase = (recs[:, -1].astype(np.int64) - (recs[:, :-1].astype(np.int64) * key).sum(axis=1)).astype(np.uint32).view(np.int32)
    assert int("".join("1" if v > 0 else "0" for v in phase), 2) == 9876          # test/bootstrap_modules/plain1.txt
    assert not np.array_equal(keygen.lwe_key_from_seed([100, 20032, 21342], 630), ref)
