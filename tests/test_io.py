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
