"""libtfhe-compatible ciphertext files (the format the reference's programs exchange, e.g. test/bootstrap_modules/*.data).

`export_gate_bootstrapping_ciphertext_toFile` writes, per LweSample, little-endian
    int32 type_uid = 42 | int32 a[n] | int32 b | float64 current_variance
(src/bootstrap_modules.cpp:100-104 writes 32 of them per file).  Key files (secret / cloud key sets)
use libtfhe-internal layouts that are not documented in the reference tree and are not handled here.
"""
import numpy as np

LWE_SAMPLE_TYPE_UID = 42


def read_ciphertexts(path, n):
    """-> (records int32[count][n+1], variances float64[count])."""
    raw = np.fromfile(path, dtype=np.uint8)
    rec = 4 + 4 * (n + 1) + 8
    if raw.size % rec:
        raise ValueError(f"{path}: size {raw.size} is not a multiple of the {rec}-byte LweSample record for n = {n}")
    raw = raw.reshape(-1, rec)
    uid = raw[:, :4].copy().view("<i4").ravel()
    if not np.all(uid == LWE_SAMPLE_TYPE_UID):
        raise ValueError(f"{path}: unexpected type_uid {set(uid.tolist())} (expected 42)")
    words = raw[:, 4:4 + 4 * (n + 1)].copy().view("<i4")
    var = raw[:, 4 + 4 * (n + 1):].copy().view("<f8").ravel()
    return np.ascontiguousarray(words, np.int32), var


def write_ciphertexts(path, records, variances=None):
    records = np.ascontiguousarray(records, np.int32)
    count, words = records.shape
    var = np.zeros(count) if variances is None else np.asarray(variances, np.float64)
    out = np.empty((count, 4 + 4 * words + 8), np.uint8)
    out[:, :4] = np.full(count, LWE_SAMPLE_TYPE_UID, "<i4").view(np.uint8).reshape(count, 4)
    out[:, 4:4 + 4 * words] = records.astype("<i4").view(np.uint8).reshape(count, 4 * words)
    out[:, 4 + 4 * words:] = var.astype("<f8").view(np.uint8).reshape(count, 8)
    out.tofile(path)
