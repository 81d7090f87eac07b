import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "torus-fhe_amd")):
    if d not in sys.path:
        sys.path.insert(0, d)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Build the native pieces in-tree when they are missing or stale (hipcc cross-compiles gfx950 without a GPU)."""
    import subprocess
    lib = os.path.join(ROOT, "torus-fhe_amd", "lib", "libthfhe_hip.so")
    csrc = os.path.join(ROOT, "torus-fhe_amd", "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(f) for f in srcs):
        if os.path.exists("/opt/rocm/bin/hipcc"):
            subprocess.run(["make", "-s", "-C", csrc], check=True)


@pytest.fixture(scope="session")
def O():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def sk128(O):
    """SK-128 key material for the reference's fixture LWE key (seed {100,20032,21341}), oracle keygen."""
    p = O.make_params("SK-128")
    s = O.SIGMAS["SK-128"]
    K = O.SKKeys(p, 0x5EED0001, s["bk"], s["ks"], lwe_key=O.fixture_key())
    return p, K, O.Oracle(p, K.bk, K.ksk)


@pytest.fixture(scope="session")
def sk_small(O):
    """Reduced-n single-key set (n=16) for fast schoolbook cross-checks."""
    p = O.make_params("SK-128", n=16)
    K = O.SKKeys(p, 77, 2.0**-25, 2.0**-15)
    return p, K, O.Oracle(p, K.bk, K.ksk)


def full_adder(multi, a, b, carry_in):
    """The reference's ripple adder wiring (src/bootstrap_modules.cpp:20-44) on MSB-first bit arrays.
    multi(jobs) evaluates a list of independent (op, x, y) batches and returns their outputs in order;
    returns (sum[nb], carry[nb]) with carry[nb-1] = carry_in."""
    import oracle_lib as OL
    nb = a.shape[0]
    sum1, carry1 = multi([(OL.XOR, a, b), (OL.AND, a, b)])
    sum2 = np.zeros_like(a)
    carry = np.zeros_like(a)
    carry[nb - 1] = carry_in
    for i in range(nb - 1, -1, -1):
        s, c2 = multi([(OL.XOR, sum1[i:i + 1], carry[i:i + 1]), (OL.AND, sum1[i:i + 1], carry[i:i + 1])])
        sum2[i] = s[0]
        if i != 0:
            carry[i - 1] = multi([(OL.OR, carry1[i:i + 1], c2)])[0][0]
    return sum2, carry


def threaded_multi(gates):
    """multi() for the CPU oracle: independent jobs run on Python threads (ctypes drops the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(4)

    def multi(jobs):
        return list(pool.map(lambda j: gates(*j), jobs))
    return multi
