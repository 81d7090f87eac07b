"""The context cache + call combiner behind the libtfhe-named entry points (torus-fhe_amd/csrc/thfhe_keyslot.h, used by tfhe_shim.cpp)
under ThreadSanitizer on the CPU: eight caller threads while the key set at the address is swapped / forgotten.  GPU twin:
tests/test_gpu_tfhe_shim.py::test_key_swap_at_one_address_under_concurrent_callers."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_keyslot_cache_under_tsan_with_key_swaps():
    d = os.path.join(ROOT, "tests", "cpp")
    subprocess.run(["make", "-s", "-C", d, "keyslot_test"], check=True)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    r = subprocess.run([os.path.join(d, "keyslot_test")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["errors"] == 0 and res["built"] == res["destroyed"] > 1 and res["slots_left"] == 0
    assert res["launches"] < res["gates"]      # concurrent calls were combined into shared launches
