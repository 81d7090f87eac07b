"""sha256 over the engine's kernel sources: ties a committed counter file (profiles/counters_*.json) to the code it was measured on.
bench.py refuses counters whose hash differs from the tree it runs from (roofline.traffic -> null)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "torus-fhe_amd", "csrc")


def kernel_source_hash():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h", ".cpp")) or name == "Makefile":
            h.update(name.encode() + b"\0")
            h.update(open(os.path.join(CSRC, name), "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(kernel_source_hash())
